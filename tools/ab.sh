#!/bin/bash
# Same-box A/B of two builds of librt_amd.so: tools/ab.sh libA.so libB.so [rounds] — alternates the headline bench between them
# (boxes differ by a few percent in clocks, so timings from different gpurun calls do not compare at the 2 % level).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for lib in $A $B; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c4 $BENCH_ARGS 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', d['config']['kernel'], 'ms/step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'], 'sha', d['config']['frame_sha256_16'])" || exit 1
  done
done
