#!/bin/bash
# same-box A/B of two libraries on headline / C4 / C5: tools/lab/r3/r3_ab2.sh libA.so libB.so [tag]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=$1; B=$2; TAG=${3:-ab2}
{
echo "# headline 1920x1080 spp 64 depth 4"
tools/abn.sh 3 $A $B
echo "# C4 3840x2160 spp 256 depth 8"
BENCH_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" tools/abn.sh 2 $A $B
echo "# C5 region 1024 3840x2160 spp 1024 depth 8"
ABN_TIMEOUT=420 BENCH_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" tools/abn.sh 1 $A $B
} 2>&1 | tee gpurun_out/r3_$TAG.txt
