#!/bin/bash
# RT_PL_STREAM matrix: 0 never, 1 streaming stores (k_paths), 2 streaming loads (k_accumulate_paths), 3 both, unset = by size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r3_pl_stream.txt; : > $O
for args in "" "--share-of 0/8" "--spp 4" "--spp 16" "--spp 32" "--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1"; do
  echo "## $args" >> $O
  for i in 1 2; do for m in 0 1 2 3 auto; do
    if [ $m = auto ]; then unset RT_PL_STREAM; else export RT_PL_STREAM=$m; fi
    echo -n "stream=$m " >> $O
    BENCH_ARGS="$args" ABN_TIMEOUT=200 tools/abn.sh 1 raytrace_amd/librt_amd.so >> $O 2>&1 || exit 1
  done; done
done
cat $O
