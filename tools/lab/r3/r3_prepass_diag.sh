#!/bin/bash
# what bounds the prepass: the shipped k_primary2 (one atomic per workgroup and round, two barriers) / one returning atomic per
# wave and tile, no barriers / no append at all (wrong worklist, timing only) — prepass-only frames (depth 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=raytrace_amd
{
for size in "--spp 1 --depth 0 --steps 50 --warmup 5" "--width 1024 --height 1024 --spp 1 --depth 0 --steps 50 --warmup 5" "--width 3840 --height 2160 --spp 1 --depth 0 --steps 30 --warmup 3"; do
  echo "## $size"
  BENCH_ARGS="$size" tools/abn.sh 2 $L/librt_amd.so $L/librt_amd_pwave.so $L/librt_amd_pnoapp.so
done
} 2>&1 | tee gpurun_out/r3_prepass_diag.txt
