#!/bin/bash
# Per-launch durations of the post passes (rocprofv3 --kernel-trace --stats); args: width height repeats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/post_kstats; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/post_profile.py $* > $OUT/run.log 2>&1
echo "trace exit $?"; tail -1 $OUT/run.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/post_kstats/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "denoise" in r["Name"] or "finalize" in r["Name"] or "persist" in r["Name"] or "primary" in r["Name"]:
            print(r["Name"][:50].ljust(52), r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3, "max_us", float(r["MaxNs"]) / 1e3)
PY
