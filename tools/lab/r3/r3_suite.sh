#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, rt_upload_slice's wall time per call, the VALU issue-cost micro-benchmark.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/slab_walltime.py > gpurun_out/r3_slab_walltime.jsonl 2> gpurun_out/slab_walltime.err; echo "slab walltime exit $?"; cat gpurun_out/r3_slab_walltime.jsonl | cut -c1-260
echo "# tools/ubench/valu_rate (gfx950, shader cycles per wave64 instruction and SIMD; 1 / 2 / 4 waves per SIMD)" > gpurun_out/r3_ubench_valu_rate.txt
timeout -k 10 100 tools/ubench/valu_rate >> gpurun_out/r3_ubench_valu_rate.txt 2>&1; echo "valu_rate exit $?"
echo "# tools/ubench/swap_rate" >> gpurun_out/r3_ubench_valu_rate.txt
timeout -k 10 60 tools/ubench/swap_rate >> gpurun_out/r3_ubench_valu_rate.txt 2>&1
tail -14 gpurun_out/r3_ubench_valu_rate.txt
