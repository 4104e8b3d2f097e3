#!/bin/bash
# Same-box comparison of several builds: tools/abn.sh ROUNDS lib1.so lib2.so ... (see tools/ab.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=$1; shift
for i in $(seq $N); do
  for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 ${ABN_TIMEOUT:-200} python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c4 $BENCH_ARGS 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', d['config']['kernel'], 'ms/step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'], 'sha', d['config']['frame_sha256_16'])" || exit 1
  done
done
