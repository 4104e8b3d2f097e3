#!/bin/bash
# round 4, GPU visit 4: C5 with / without the per-brick map, one / two lanes, 86 / 172 samples per launch; both poses
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'fif', d['config']['frames_in_flight'], 'lanes', d['config']['launches_in_flight'], 'ms/step', d['ms_per_step'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'Grays/s', round(d['value']/1e3,2), 'sha', d['config']['frame_sha256_16'])"; }
C5="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline"
for lib in librt_amd_brick.so librt_amd.so; do   # tools/variant.sh brick rt_paths.hip -DRT_PATHS_BRICK_MAP=1; export RT_BRICK_MAP=1
  for cfg in "RT_LANES=2 RT_PERSIST_LIGHT_GIB=16" "RT_LANES=1 RT_PERSIST_LIGHT_GIB=16" "RT_LANES=2 RT_PERSIST_LIGHT_GIB=32"; do
    env RT_AMD_LIB=$PWD/raytrace_amd/$lib $cfg timeout -k 10 500 python bench.py $C5 2>/dev/null | tail -1 | line "$lib $cfg c5"
  done
  env RT_AMD_LIB=$PWD/raytrace_amd/$lib timeout -k 10 500 python bench.py $C5 --pose=-120,-512,160,1.5707964,-0.3 2>gpurun_out/r4/c5t.err | tail -1 | line "$lib c5terrain"
done
tail -3 gpurun_out/r4/c5t.err
