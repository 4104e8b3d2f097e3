#!/bin/bash
# One bench.py line per BASELINE.json config on one GPU (C2, C3, C4, C5); the lines land in gpurun_out/bench_configs.jsonl
mkdir -p gpurun_out; out=gpurun_out/r3_bench_configs.jsonl; : > $out
run() { echo "# $*" >> $out; timeout -k 10 500 python bench.py "$@" 2>/dev/null | grep "^{" >> $out; tail -1 $out | cut -c1-220; }
run --spp 1 --depth 0 --steps 50 --warmup 5 --no-cpu-baseline                                                   # C2: primary rays only
run --steps 10 --warmup 2                                                                                       # C3: headline
run --width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline                       # C4 on one GPU
run --region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline        # C5 on one GPU
run --width 1024 --height 1024 --spp 1 --depth 2 --steps 50 --warmup 5 --no-cpu-baseline                        # the reference's own frame
