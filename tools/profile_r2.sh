#!/bin/bash
# Round-2 profile set on the GPU box (rocprofv3; every --pmc set in its own pass).  Raw outputs land in gpurun_out/prof_r2/;
# tools/pmc_to_json.py condenses them into profiles/r2_counters.json, the CSV/TXT summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r2; rm -rf $OUT; mkdir -p $OUT   # NB: gpurun MERGES into the local gpurun_out/ — delete the local copy first too
HEAD_ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
HEAD_STATS_ARGS="--steps 10 --warmup 3 --no-cpu-baseline"   # kernel-trace pass: enough launches that cold ones do not set the average
C5_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 1 --warmup 1 --no-cpu-baseline"
C4_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
run() {   # tag, bench args, rocprof args
  local tag=$1 args=$2; shift 2
  timeout -k 10 280 rocprofv3 "$@" --output-format csv -d $OUT/$tag -- python3 bench.py $args > $OUT/$tag.log 2>&1
  echo "$tag exit $?"
}
for w in head c5 c4; do
  case $w in head) A=$HEAD_ARGS;; c5) A=$C5_ARGS;; c4) A=$C4_ARGS;; esac
  if [ $w = head ]; then run ${w}_stats "$HEAD_STATS_ARGS" --kernel-trace --stats; else run ${w}_stats "$A" --kernel-trace --stats; fi
  run ${w}_fetch "$A" --pmc FETCH_SIZE
  run ${w}_write "$A" --pmc WRITE_SIZE
  run ${w}_tcc "$A" --pmc TCC_HIT_sum TCC_MISS_sum
  if [ $w != c4 ]; then
    run ${w}_sq1 "$A" --pmc $P1
    run ${w}_sq2 "$A" --pmc $P2
    run ${w}_sq3 "$A" --pmc $P3
  fi
done
# slab upload: kernel times of the incremental rt_upload_slice at R = 256 and 512
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/slab_stats -- python3 tools/slab_timing.py > $OUT/slab_stats.log 2>&1; echo "slab exit $?"
# post passes at 4K
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/post_stats -- python3 tools/post_profile.py 3840 2160 10 > $OUT/post_stats.log 2>&1; echo "post exit $?"
python3 tools/pmc_to_json.py $OUT > $OUT/r2_counters.json && cp $OUT/r2_counters.json gpurun_out/r2_counters.json
for t in head c5 c4 slab post; do f=$(find $OUT/${t}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r2_${t}_kernel_stats.csv; done
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r2_counters.json"))
for w, ks in d["workloads"].items():
    for k, r in ks.items():
        print(w, k, {x: r[x] for x in r if x != "raw"})
PY
