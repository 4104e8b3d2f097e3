#!/bin/bash
# One bench.py line per BASELINE.json config on one GPU (C2, C3, C4, C5); the lines land in gpurun_out/bench_configs.jsonl
mkdir -p gpurun_out; out=gpurun_out/r4_bench_configs.jsonl; : > $out
run() { echo "# $*" >> $out; timeout -k 10 500 python bench.py "$@" 2>/dev/null | grep "^{" >> $out; tail -1 $out | cut -c1-220; }
run --spp 1 --depth 0 --steps 50 --warmup 5 --no-cpu-baseline                                                   # C2: primary rays only
run --steps 10 --warmup 2                                                                                       # C3: headline
run --width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline                       # C4 on one GPU
run --region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline        # C5 on one GPU
run --region 1024 --width 3840 --height 2160 --spp 128 --depth 8 --steps 1 --warmup 1 --no-cpu-baseline --no-reference-frame --pose=-120,-512,160,1.5707964,-0.3   # C5, terrain-heavy pose (8.7 x the rays per sample), an eighth of the samples
run --width 1024 --height 1024 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame --frames-in-flight 1     # the reference's own frame (k_frame), one frame in flight
run --width 1024 --height 1024 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame                         # ... two
run --width 256 --height 256 --spp 1 --depth 2 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-frame --frames-in-flight 1
