#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -8 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
tools/lab/r3/r3_small.sh
tools/abn.sh 1 raytrace_amd/librt_amd.so
