#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the default bench command, per kernel (separate --pmc passes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcf; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/f.log 2>&1; echo "fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/w.log 2>&1; echo "write exit $?"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmcf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "persist" in k or "accumulate" in k: print("%-50s %-12s %.1f per dispatch (%d)" % (k[:50], c, v / n, n))
PY
