#!/bin/bash
# Per-kernel times of the default bench command (rocprofv3 --kernel-trace --stats); prints the stats CSV.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_run.log 2>&1
echo "trace exit $?"
for f in $(find $OUT -name "*kernel_stats.csv"); do cut -d, -f1-6 $f | head -14; done
