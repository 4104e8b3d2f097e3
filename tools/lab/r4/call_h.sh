#!/bin/bash
# C4 (and C5) on the final code: one lane against two, pass threshold
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'x', r['launches_per_frame'], 'spl', d['config']['samples_per_launch'], 'sha', d['config']['frame_sha256_16'])"; }
C4="--width 3840 --height 2160 --spp 256 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-frame"
C5="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-frame"
for lanes in 2 1 2 1; do RT_LANES=$lanes timeout -k 10 300 python bench.py $C4 2>/dev/null | tail -1 | line "c4 lanes=$lanes"; done
for t in 36 44 48; do RT_PERSIST_THRESHOLD=$t timeout -k 10 300 python bench.py $C4 2>/dev/null | tail -1 | line "c4 threshold=$t"; done
for lanes in 2 1; do RT_LANES=$lanes timeout -k 10 400 python bench.py $C5 2>/dev/null | tail -1 | line "c5 lanes=$lanes"; done
