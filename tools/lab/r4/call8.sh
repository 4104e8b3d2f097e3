#!/bin/bash
# round 4, GPU visit 8: is the drain slow because of the streaming (nt) light-record stores?  wave / workgroup end times with RT_PL_STREAM = default (3 for the headline) / 0 / 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for mode in "" 0 1 2; do
  echo "## RT_PL_STREAM=$mode" >> gpurun_out/r4/drain_pl_stream.txt
  for w in "" "0/8"; do
    echo "# $w" >> gpurun_out/r4/drain_pl_stream.txt
    if [ -z "$mode" ]; then RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_wt.so timeout -k 10 200 python tools/lab/r4/wg_end_times.py $w >> gpurun_out/r4/drain_pl_stream.txt 2>&1
    else RT_PL_STREAM=$mode RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_wt.so timeout -k 10 200 python tools/lab/r4/wg_end_times.py $w >> gpurun_out/r4/drain_pl_stream.txt 2>&1; fi
  done
done
grep -E "^##|^# |waves' own drain|CU idle" gpurun_out/r4/drain_pl_stream.txt
