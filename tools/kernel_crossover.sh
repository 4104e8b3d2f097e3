#!/bin/bash
# k_paths vs k_persist across workload sizes (which one should RT_KERNEL_DEFAULT run?)
cd $GRAFT_REPO_ROOT
run() { for k in paths persistent; do timeout -k 10 300 python bench.py --kernel $k --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$k', '$*', 'ms/step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'])"; done; }
run --width 1024 --height 1024 --spp 1 --depth 2 --steps 50 --warmup 5
run --spp 1 --depth 4 --steps 50 --warmup 5
run --spp 4 --depth 4 --steps 30 --warmup 5
run --spp 16 --depth 4 --steps 20 --warmup 3
run --width 256 --height 256 --spp 64 --depth 4 --steps 30 --warmup 5
run --width 3840 --height 2160 --spp 256 --depth 8 --steps 2 --warmup 1
