#!/bin/bash
# One GPU-box visit while k_paths is being reworked: the -m gpu suite on the working tree's library, then the same-box A/B of
# HEAD's rt_paths.hip (tools/variant_head.sh) against the working tree on the headline frame, C4 and C5.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -x tools/ubench/lds_align ] && timeout -k 5 30 tools/ubench/lds_align > gpurun_out/r3_lds_align.txt 2>&1 && cat gpurun_out/r3_lds_align.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -8 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
A=raytrace_amd/librt_amd_head.so; B=raytrace_amd/librt_amd.so
{
echo "# headline 1920x1080 spp 64 depth 4"
tools/abn.sh ${AB_ROUNDS:-3} $A $B
echo "# C4 3840x2160 spp 256 depth 8"
BENCH_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" tools/abn.sh 1 $A $B
echo "# C5 region 1024 3840x2160 spp 1024 depth 8"
ABN_TIMEOUT=420 BENCH_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" tools/abn.sh 1 $A $B
} 2>&1 | tee gpurun_out/r3_ab.txt
timeout -k 10 200 python tools/slab_walltime.py > gpurun_out/r3_slab_walltime.jsonl 2> gpurun_out/slab_walltime.err; echo "slab walltime exit $?"; cut -c1-230 gpurun_out/r3_slab_walltime.jsonl
