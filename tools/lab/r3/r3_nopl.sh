#!/bin/bash
# what would removing the path-light round trip buy?  Same-box A/B of the shipped library against a diagnostic
# build whose k_paths never stores its light records (tools/variant.sh nopl rt_paths.hip -DRT_DIAG_NO_PL_STORE: wrong frames,
# timing only) on the headline frame, C4 and C5.  Also records the VALU issue-cost micro-benchmark's output.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=raytrace_amd/librt_amd.so; B=raytrace_amd/librt_amd_nopl.so
{
echo "# headline 1920x1080 spp 64 depth 4"
tools/abn.sh 3 $A $B
echo "# C4 3840x2160 spp 256 depth 8"
BENCH_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" tools/abn.sh 2 $A $B
echo "# C5 region 1024 3840x2160 spp 1024 depth 8"
ABN_TIMEOUT=420 BENCH_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" tools/abn.sh 1 $A $B
} 2>&1 | tee gpurun_out/r3_nopl_ab.txt
echo "# tools/ubench/valu_rate (gfx950, shader cycles per wave64 instruction and SIMD; 1 / 2 / 4 waves per SIMD)" > gpurun_out/r3_ubench_valu_rate.txt
timeout -k 10 120 tools/ubench/valu_rate >> gpurun_out/r3_ubench_valu_rate.txt 2>&1
echo "# tools/ubench/swap_rate" >> gpurun_out/r3_ubench_valu_rate.txt
timeout -k 10 120 tools/ubench/swap_rate >> gpurun_out/r3_ubench_valu_rate.txt 2>&1
tail -5 gpurun_out/r3_ubench_valu_rate.txt
