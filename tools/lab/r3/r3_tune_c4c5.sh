#!/bin/bash
# parked-lane threshold and chunk size on C4 and C5 (round 3 kernel)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=raytrace_amd/librt_amd.so
{
for th in 28 32 36 40 44; do
  echo "## C4 threshold $th"; RT_PERSIST_THRESHOLD=$th BENCH_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" tools/abn.sh 1 $L
done
for th in 28 32 36 40 44; do
  echo "## C5 threshold $th"; RT_PERSIST_THRESHOLD=$th ABN_TIMEOUT=420 BENCH_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" tools/abn.sh 1 $L
done
for ch in 64 128 256; do
  echo "## headline chunk $ch"; RT_PERSIST_CHUNK=$ch tools/abn.sh 2 $L
done
} 2>&1 | tee gpurun_out/r3_tune_c4c5.txt
