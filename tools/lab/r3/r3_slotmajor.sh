#!/bin/bash
# k_paths' path order: samples of one pixel consecutively (RT_PATHS_SLOT_MAJOR=1) against slots of one sample consecutively
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RT_PATHS_SLOT_MAJOR=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q > gpurun_out/pytest_slotmajor.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_slotmajor.log
[ $rc -ne 0 ] && exit $rc
L=raytrace_amd/librt_amd.so
{
for cfg in "" "--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1" "--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1" "--spp 4 --depth 4 --steps 50 --warmup 5" "--spp 16 --depth 4"; do
  echo "## $cfg"
  for sm in 0 1 0 1; do echo "slot_major $sm"; RT_PATHS_SLOT_MAJOR=$sm ABN_TIMEOUT=420 BENCH_ARGS="$cfg" tools/abn.sh 1 $L; done
done
} 2>&1 | tee gpurun_out/r3_slotmajor.txt
