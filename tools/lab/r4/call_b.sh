#!/bin/bash
# PMC passes of k_frame on the reference's frame (each --pmc set in its own process; program directly after `--`)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_frame; rm -rf $OUT; mkdir -p $OUT
A="tools/lab/r4/frame_loop.py 1024 1024 1 2 7 8"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_IFETCH"
P3="SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU"
P4="TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $A > $OUT/stats.log 2>&1; echo "stats $?"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  timeout -k 10 120 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $A > $OUT/p$i.log 2>&1; echo "p$i $?"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_frame/p*")):
    if not d[-1].isdigit(): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, cs in acc.items():
        if "k_frame" in k or "k_persist" in k or "k_primary" in k:
            print(d[-2:], k, {c: round(v / max(n[(k, c)], 1)) for c, v in cs.items()})
for f in glob.glob("gpurun_out/prof_frame/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
