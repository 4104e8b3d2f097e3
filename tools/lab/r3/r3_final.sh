#!/bin/bash
# Final visit of a round: the whole GPU suite, then the profile set for the workloads named in PROFILE_WORKLOADS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -6 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
tools/profile_r3.sh
