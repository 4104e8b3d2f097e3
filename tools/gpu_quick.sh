#!/bin/bash
# Quick GPU visit: the parity tests that exercise the default path kernel, then the headline bench (HIP-event time per launch).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "paths or PATHS or edge or scheduling or fuzz or random or batches or tile_split or primary_cache or tables" > gpurun_out/pytest_quick.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_quick.log
[ $rc -ne 0 ] && exit $rc
for extra in "$@"; do
  env $extra timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$extra', d['config']['kernel'], 'ms/step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'], 'sha', d['config']['frame_sha256_16'])"
done
