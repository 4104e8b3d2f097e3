#!/bin/bash
# round 4, GPU visit 1: baseline bench, two frames in flight emulated with two contexts, workgroup end times of k_paths
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4/bench_base.log 2>&1 && tail -1 gpurun_out/r4/bench_base.log | cut -c1-400
timeout -k 10 400 python tools/lab/r4/two_ctx_overlap.py headline share8 spp4 small > gpurun_out/r4/two_ctx.jsonl 2> gpurun_out/r4/two_ctx.err && cat gpurun_out/r4/two_ctx.jsonl
for w in "" "0/8" "0/1 1920 1080 4 4"; do
  echo "# $w" >> gpurun_out/r4/wg_end.txt
  RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_wt.so timeout -k 10 200 python tools/lab/r4/wg_end_times.py $w >> gpurun_out/r4/wg_end.txt 2>&1
done
cat gpurun_out/r4/wg_end.txt
