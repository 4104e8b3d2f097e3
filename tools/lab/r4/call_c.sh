#!/bin/bash
cd $GRAFT_REPO_ROOT
T=tools/lab/r4/frame_time.py
for t in 16 24 48; do RT_FRAME_THRESHOLD=$t timeout -k 10 100 python3 $T 7 1024 1024 1 2; done
for k in 5 6 8; do RT_FRAME_TILES=$k timeout -k 10 100 python3 $T 7 1024 1024 1 2 1920 1080 1 2; done
echo "--- multi-sample small frames: frame / persistent / paths"
C="256 256 4 2 256 256 16 4 256 256 64 4 512 512 4 4 512 512 16 4 1024 1024 2 4 1920 1080 2 4"
for kern in 7 3 5; do timeout -k 10 200 python3 $T $kern $C; done
