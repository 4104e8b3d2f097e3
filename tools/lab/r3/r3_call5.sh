#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -8 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
L=raytrace_amd
{
echo "# headline: steps per look / shadow repetitions x parked-lane threshold"
for th in 36 40 44; do
  echo "threshold $th"
  RT_PERSIST_THRESHOLD=$th tools/abn.sh 1 $L/librt_amd.so $L/librt_amd_c4r7.so $L/librt_amd_c4rB.so $L/librt_amd_c5r17.so $L/librt_amd_c5rF.so
done
echo "# HEAD for reference"
tools/abn.sh 1 $L/librt_amd_head.so
} 2>&1 | tee gpurun_out/r3_variants2.txt
# the reference's own frame: per-kernel times and gaps
OUT=gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --width 1024 --height 1024 --spp 1 --depth 2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/run.log 2>&1; echo "small exit $?"
tail -1 $OUT/run.log | cut -c1-300
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat $f | cut -c1-160
