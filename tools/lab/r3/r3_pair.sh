#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=${1:-pair}
RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_$V.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q > gpurun_out/pytest_$V.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_$V.log
[ $rc -ne 0 ] && exit $rc
tools/lab/r3/r3_ab2.sh raytrace_amd/librt_amd.so raytrace_amd/librt_amd_$V.so $V
