#!/bin/bash
# same-box A/B of $VARIANTS (libraries under raytrace_amd/) on headline / C4 / C5 / a scrolled window: tools/lab/r3/r3_ab3.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r3_ab3.txt; : > $O
for args in "" "--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" "--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" "--lr 48,-32,16"; do
  echo "## $args" >> $O
  BENCH_ARGS="$args" ABN_TIMEOUT=200 tools/abn.sh 2 raytrace_amd/librt_amd.so $VARIANTS >> $O 2>&1 || exit 1
done
cat $O
