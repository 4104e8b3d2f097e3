#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel', d['config']['kernel'], 'fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'frac', d['roofline']['frac'], 'sha', d['config']['frame_sha256_16'])"; }
for wh in "1024 1024" "256 256"; do set -- $wh
for fif in 1 2; do
  timeout -k 10 200 python bench.py --width $1 --height $2 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "ref $1x$2 fif=$fif"
done; done | tee gpurun_out/r4/ref_frame_bench.txt
for post in "" "--post"; do ./raytrace_amd/rt_bench --frames 600 $post | tail -2 | head -1; done
