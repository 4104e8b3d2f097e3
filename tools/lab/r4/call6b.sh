#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_post_passes.py tests/test_streaming.py -m gpu -x -q -k "bench_gather or gather_gbuffer or rt_bench or cameras_outside or post or streaming" > gpurun_out/r4/pytest_gpu_b.log 2>&1; rc=$?
tail -6 gpurun_out/r4/pytest_gpu_b.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; tail -1 gpurun_out/r4/bench_default.json | cut -c1-2500
tools/lab/r4/call7.sh
