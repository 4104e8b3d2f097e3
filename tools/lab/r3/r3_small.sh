#!/bin/bash
# Small launches: cached primaries (k_primary2 + path kernel) against the un-cached path kernel that traces its own primary rays
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', '->', d['config']['kernel'], 'ms/step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'])"; }
{
for size in "--width 1024 --height 1024 --spp 1 --depth 2" "--spp 1 --depth 4" "--spp 2 --depth 4" "--spp 4 --depth 4" "--width 256 --height 256 --spp 1 --depth 2"; do
  for mode in "" "--no-cache-primary" "--kernel paths"; do
    run $size --steps 50 --warmup 5 $mode
  done
done
} 2>&1 | tee gpurun_out/r3_small.txt
