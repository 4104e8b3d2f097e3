#!/bin/bash
# headline with two frames in flight across the pass threshold and the chunk size of k_paths (tuned in round 3 with one launch in flight)
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'sha', d['config']['frame_sha256_16'])"; }
for t in 32 36 40 44; do RT_PERSIST_THRESHOLD=$t timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-frame 2>/dev/null | tail -1 | line "threshold $t"; done
for c in 64 192 256; do RT_PERSIST_CHUNK=$c timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-frame 2>/dev/null | tail -1 | line "chunk $c"; done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-frame 2>/dev/null | tail -1 | line "default"
