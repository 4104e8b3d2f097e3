#!/bin/bash
# Round-4 profile set on the GPU box (rocprofv3; every --pmc set in its own pass; the program goes directly after `--`).  Raw
# outputs land in gpurun_out/prof_r4/; tools/pmc_to_json.py condenses them into r4_counters.json (copied to profiles/ by hand
# together with the kernel-stats CSVs).  Run tools/isa_hist.py first: the counters' pipe_busy_weighted uses its issue cost.
# The kernel-trace pass of the headline runs TWICE: as bench.py runs by default (two frames in flight: a launch's begin-to-end time
# includes its start on the CUs the launch in front of it is leaving) and with one frame in flight.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r4; rm -rf $OUT; mkdir -p $OUT   # NB: gpurun MERGES into the local gpurun_out/ — delete the local copy first too
X="--no-cpu-baseline --no-reference-frame"
HEAD_ARGS="--steps 3 --warmup 1 $X"
HEAD_STATS_ARGS="--steps 10 --warmup 3 $X"   # kernel-trace pass: enough launches that cold ones do not set the average
C5_ARGS="--region 1024 --width 3840 --height 2160 --spp 1024 --depth 8 --steps 1 --warmup 1 $X"
C5T_ARGS="$C5_ARGS --pose=-120,-512,160,1.5707964,-0.3 --spp 128"     # the terrain-heavy pose (8.7 x the rays per sample): an eighth of the samples
REF_ARGS="--width 1024 --height 1024 --spp 1 --depth 2 --steps 50 --warmup 5 $X"    # the reference's own frame: one launch of k_frame
C4_ARGS="--width 3840 --height 2160 --spp 256 --depth 8 --steps 2 --warmup 1 $X"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
run() {   # tag, bench args, rocprof args
  local tag=$1 args=$2; shift 2
  timeout -k 10 280 rocprofv3 "$@" --output-format csv -d $OUT/$tag -- python3 bench.py $args > $OUT/$tag.log 2>&1
  echo "$tag exit $?"
}
for w in ${PROFILE_WORKLOADS:-head ref c4 c5 c5t}; do
  case $w in head) A=$HEAD_ARGS;; c5) A=$C5_ARGS;; c5t) A=$C5T_ARGS;; c4) A=$C4_ARGS;; ref) A=$REF_ARGS;; esac
  if [ $w = head ]; then
    run ${w}_stats "$HEAD_STATS_ARGS" --kernel-trace --stats
    run ${w}1_stats "$HEAD_STATS_ARGS --frames-in-flight 1" --kernel-trace --stats
  else run ${w}_stats "$A" --kernel-trace --stats; fi
  run ${w}_fetch "$A" --pmc FETCH_SIZE
  run ${w}_write "$A" --pmc WRITE_SIZE
  run ${w}_tcc "$A" --pmc TCC_HIT_sum TCC_MISS_sum
  if [ $w != c5t ]; then
    run ${w}_sq1 "$A" --pmc $P1
    run ${w}_sq2 "$A" --pmc $P2
    run ${w}_sq3 "$A" --pmc $P3
  fi
done
if [ -z "$PROFILE_WORKLOADS" ] || [[ "$PROFILE_WORKLOADS" == *post* ]]; then
  # slab upload: kernel times of the incremental rt_upload_slice at R = 256 and 512
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/slab_stats -- python3 tools/slab_timing.py > $OUT/slab_stats.log 2>&1; echo "slab exit $?"
  # post passes at 4K
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/post_stats -- python3 tools/post_profile.py 3840 2160 10 > $OUT/post_stats.log 2>&1; echo "post exit $?"
fi
python3 tools/pmc_to_json.py $OUT > $OUT/r4_counters.json && cp $OUT/r4_counters.json gpurun_out/r4_counters.json
for t in head head1 ref c5 c5t c4 slab post; do f=$(find $OUT/${t}_stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_${t}_kernel_stats.csv; done
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_counters.json"))
for w, ks in d["workloads"].items():
    for k, r in ks.items():
        print(w, k, {x: r[x] for x in r if x != "raw"})
PY
