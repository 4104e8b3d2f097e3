#!/bin/bash
# round 4, GPU visit 7: C4 across light-record budgets with one / two lanes, scale emulation with two frames in flight, small frames, slab wall times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'fif', d['config']['frames_in_flight'], 'lanes', d['config']['launches_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'MB', d['config']['context_device_bytes']>>20, 'sha', d['config']['frame_sha256_16'])"; }
C4="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline"
for gib in 2 4 8 16 32; do
  for lanes in 2 1; do
    env RT_LANES=$lanes RT_PERSIST_LIGHT_GIB=$gib timeout -k 10 300 python bench.py $C4 2>/dev/null | tail -1 | line "c4 gib=$gib lanes=$lanes"
  done
done | tee gpurun_out/r4/c4_budgets.txt
SCALE_FIF=2 timeout -k 10 400 python tools/scale_emulation.py headline c4 > gpurun_out/r4/scale_emulation_fif2.jsonl 2> gpurun_out/r4/scale2.err; python -c "
import json
for l in open('gpurun_out/r4/scale_emulation_fif2.jsonl'):
    d=json.loads(l); print(d['workload'], {n: (v['slowest_ms'], v['render_speedup_vs_1']) for n, v in d['n'].items()})"
SCALE_FIF=1 timeout -k 10 400 python tools/scale_emulation.py headline > gpurun_out/r4/scale_emulation_fif1.jsonl 2> gpurun_out/r4/scale1.err; python -c "
import json
for l in open('gpurun_out/r4/scale_emulation_fif1.jsonl'):
    d=json.loads(l); print(d['workload'], {n: (v['slowest_ms'], v['render_speedup_vs_1']) for n, v in d['n'].items()})"
for wh in "1024 1024" "256 256" "1920 1080"; do set -- $wh
  for fif in 2 1; do
    timeout -k 10 200 python bench.py --width $1 --height $2 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "small $1x$2"
  done
done | tee gpurun_out/r4/small_frames.txt
timeout -k 10 300 python tools/slab_walltime.py > gpurun_out/r4/slab_walltime.jsonl 2> gpurun_out/r4/slab.err; grep -c . gpurun_out/r4/slab_walltime.jsonl; grep "three slabs" gpurun_out/r4/slab_walltime.jsonl
for post in "" "--post"; do ./raytrace_amd/rt_bench --frames 600 $post | tail -2; done | tee gpurun_out/r4/rt_bench_frames.txt
