#!/bin/bash
# round 4, GPU visit 2: the new overlap tests, then the headline bench with one and two frames in flight, share8 emulation
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py -m gpu -x -q > gpurun_out/r4/pytest_overlap.log 2>&1; rc=$?
tail -15 gpurun_out/r4/pytest_overlap.log
[ $rc -ne 0 ] && exit $rc
for fif in 2 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight $fif > gpurun_out/r4/bench_fif$fif.log 2>&1 && tail -1 gpurun_out/r4/bench_fif$fif.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'sha', d['config']['frame_sha256_16'], d.get('reference_frame'))"
done
RT_LANES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight 1 --no-reference-frame 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lanes1 fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'])"
for fif in 2 1; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --frames-in-flight $fif --share-of 0/8 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('share 0/8 fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'])"
done
