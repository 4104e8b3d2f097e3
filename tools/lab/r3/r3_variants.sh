#!/bin/bash
# Same-box comparison of step-loop variants of k_paths (tools/variant.sh builds them) and of the parked-lane threshold, then the
# three SQ counter passes on the headline frame for the working tree's library.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=raytrace_amd
{
echo "# headline: HEAD / working tree / variants"
tools/abn.sh 2 $L/librt_amd_head.so $L/librt_amd.so $L/librt_amd_asel.so $L/librt_amd_tesel.so $L/librt_amd_chk4.so $L/librt_amd_chk2.so $L/librt_amd_chk3all.so
echo "# headline: parked-lane threshold (RT_PERSIST_THRESHOLD) on the working tree"
for th in 24 28 32 36 40 48; do echo "threshold $th"; RT_PERSIST_THRESHOLD=$th tools/abn.sh 1 $L/librt_amd.so; done
} 2>&1 | tee gpurun_out/r3_variants.txt
OUT=gpurun_out/prof_r3q; rm -rf $OUT; mkdir -p $OUT
A="--steps 3 --warmup 1 --no-cpu-baseline"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/head_stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/head_stats.log 2>&1; echo "stats exit $?"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 280 rocprofv3 --pmc $P --output-format csv -d $OUT/head_sq$i -- python3 bench.py $A > $OUT/head_sq$i.log 2>&1; echo "sq$i exit $?"
  i=$((i+1))
done
python3 tools/pmc_to_json.py $OUT > gpurun_out/r3q_counters.json 2> gpurun_out/r3q_counters.err; echo "pmc_to_json exit $?"
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3q_counters.json"))
    for w, ks in d["workloads"].items():
        for k, r in ks.items():
            print(w, k, {x: r[x] for x in r if x != "raw"})
except Exception as e:
    print("no summary:", e)
PY
