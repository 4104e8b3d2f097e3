#!/bin/bash
# Profiles the default bench command on the GPU box; results land in gpurun_out/prof/ (copy summaries to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_run.log 2>&1
echo "trace exit $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_run.log 2>&1
echo "fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_run.log 2>&1
echo "write exit $?"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc_run.log 2>&1
echo "tcc exit $?"
find $OUT -name "*.csv" | head -20
python3 - <<'PY'
import csv, glob, collections, json
out = "gpurun_out/prof"
# kernel stats
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read())
res = {}
for name, pat in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write"), ("TCC", "pmc_tcc")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/" + pat + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print("%-50s %-12s sum %.0f over %d dispatches -> %.1f per dispatch" % (k[:50], c, v, n, v / n))
        res["%s|%s" % (k, c)] = [v, n]
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
PY
tail -2 $OUT/trace_run.log
