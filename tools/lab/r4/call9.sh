#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for w in "" "0/8" "0/1 1024 1024 1 2"; do
  echo "# $w" >> gpurun_out/r4/step_times.txt
  RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_st.so timeout -k 10 200 python tools/lab/r4/step_times.py $w >> gpurun_out/r4/step_times.txt 2>&1
done
cat gpurun_out/r4/step_times.txt
C5="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline"
for gib in 8 16; do RT_PERSIST_LIGHT_GIB=$gib timeout -k 10 500 python bench.py $C5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c5 gib=$gib', 'ms/step', d['ms_per_step'], 'x', d['roofline']['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'MB', d['config']['context_device_bytes']>>20)"; done
