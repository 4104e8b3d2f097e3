#!/bin/bash
# One GPU-box visit: GPU parity suite, then the headline bench on the default kernel and on k_persist for comparison.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_paths.log 2>&1 && tail -1 gpurun_out/bench_paths.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel persistent > gpurun_out/bench_persist.log 2>&1 && tail -1 gpurun_out/bench_persist.log
