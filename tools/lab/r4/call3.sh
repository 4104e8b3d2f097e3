#!/bin/bash
# round 4, GPU visit 3: remaining overlap tests, benches with one / two frames in flight, share-of-8, brick map on C5 (fetch levels, parity, timing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py -m gpu -x -q -k "default_kernel or mirror or rt_bench" > gpurun_out/r4/pytest_overlap2.log 2>&1; rc=$?
tail -5 gpurun_out/r4/pytest_overlap2.log
[ $rc -ne 0 ] && exit $rc
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', d['roofline']['avg_launch_ms'], 'x', d['roofline']['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'sha', d['config']['frame_sha256_16'], d.get('reference_frame'))"; }
for fif in 2 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight $fif 2> gpurun_out/r4/bench_fif$fif.err | tail -1 | tee gpurun_out/r4/bench_fif$fif.json | line head
done
RT_LANES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight 1 --no-reference-frame 2>/dev/null | tail -1 | line lanes1
for fif in 2 1; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --frames-in-flight $fif --share-of 0/8 2>/dev/null | tail -1 | line share0/8
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "region_512 or region_1024 or deep_paths or c5_frame or upload_slice or scrolled_region_512" > gpurun_out/r4/pytest_brick.log 2>&1; rc=$?
tail -5 gpurun_out/r4/pytest_brick.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/fetch_levels.py > gpurun_out/r4/fetch_levels.jsonl 2> gpurun_out/r4/fetch_levels.err; cat gpurun_out/r4/fetch_levels.jsonl
C5="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 500 python bench.py $C5 2>/dev/null | tail -1 | line c5
timeout -k 10 500 python bench.py $C5 --pose -120,-512,160,1.5707964,-0.3 2>/dev/null | tail -1 | line c5terrain
