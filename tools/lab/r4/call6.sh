#!/bin/bash
# round 4, GPU visit 6: the whole GPU suite, then the bench line as the driver runs it, slab wall times, rt_bench --post
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/pytest_gpu.log 2>&1; rc=$?
tail -6 gpurun_out/r4/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; tail -1 gpurun_out/r4/bench_default.json | cut -c1-1500
