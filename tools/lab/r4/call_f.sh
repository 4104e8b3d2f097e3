#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel', d['config']['kernel'], 'fif', d['config']['frames_in_flight'], 'ms/step', d['ms_per_step'], 'latency', d['config']['latency_ms_one_frame'], 'launch', r['avg_launch_ms'], 'x', r['launches_per_frame'], 'frac', r['frac'], 'alone', (r.get('one_launch_in_flight') or {}).get('frac'), 'sha', d['config']['frame_sha256_16'])"; }
for i in 1 2; do for fif in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "head fif=$fif"
done; done
for fif in 1 2; do
  timeout -k 10 300 python bench.py --width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif 2>/dev/null | tail -1 | line "c4 fif=$fif"
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-frame --frames-in-flight $fif --share-of 0/8 2>/dev/null | tail -1 | line "share0/8 fif=$fif"
done
