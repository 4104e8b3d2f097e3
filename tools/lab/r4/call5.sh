#!/bin/bash
# round 4, GPU visit 5: tail generations — parity, then headline / share-of-8 / C4 with and without them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py tests/test_golden.py -m gpu -x -q -k "tail or golden or two_frames or lanes" > gpurun_out/r4/pytest_tail.log 2>&1; rc=$?
tail -15 gpurun_out/r4/pytest_tail.log
[ $rc -ne 0 ] && exit $rc
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'sha', d['config']['frame_sha256_16'])"; }
for cfg in "RT_TAIL_GENERATIONS=0" "RT_TAIL_GENERATIONS=5" "RT_TAIL_GENERATIONS=5 RT_TAIL_THRESHOLD=96" "RT_TAIL_GENERATIONS=3 RT_TAIL_THRESHOLD=32" "RT_TAIL_GENERATIONS=8 RT_TAIL_THRESHOLD=80"; do
  for fif in 2 1; do
    env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "head $cfg"
  done
  env $cfg timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --frames-in-flight 2 --share-of 0/8 2>/dev/null | tail -1 | line "share0/8 $cfg"
done
