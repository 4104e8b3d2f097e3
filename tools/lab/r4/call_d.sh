#!/bin/bash
# round 4, later visit: light-record budget 4 GiB against 16 GiB on C4 and C5 (two lanes), and bench.py on the reference's frame (k_frame)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel', d['config']['kernel'], 'fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'frac', d['roofline']['frac'], 'spl', d['config']['samples_per_launch'], 'MB', d['config']['context_device_bytes']>>20, 'sha', d['config']['frame_sha256_16'])"; }
C4="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-frame"
C5="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-frame"
for gib in 4 16; do
  env RT_PERSIST_LIGHT_GIB=$gib timeout -k 10 300 python bench.py $C4 2>/dev/null | tail -1 | line "c4 gib=$gib"
  env RT_PERSIST_LIGHT_GIB=$gib timeout -k 10 400 python bench.py $C5 2>/dev/null | tail -1 | line "c5 gib=$gib"
done | tee gpurun_out/r4/light_budget_4_16.txt
for fif in 1 2; do
  timeout -k 10 200 python bench.py --width 1024 --height 1024 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "ref fif=$fif"
done | tee gpurun_out/r4/ref_frame_bench.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_head.log 2>&1; tail -1 gpurun_out/r4/bench_head.log | cut -c1-3000
for post in "" "--post"; do ./raytrace_amd/rt_bench --frames 600 $post | tail -2; done | tee gpurun_out/r4/rt_bench_frames.txt
