#!/bin/bash
# usage: tools/pmc_mem.sh <tag> "<bench args>" — vector-memory pipeline counters (TA / TCP / UTCL1) of the bench command
TAG=$1; ARGS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcm_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
# NB: the TA_*, TD_* and TCP_GATE_EN*/TCP_PENDING_STALL sets never finished on this pool (each pass ran into its 240 s limit);
# only the two sets below return.
for P in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > $OUT/run$i.log 2>&1
  echo "pass $i exit $?"
done
python3 tools/pmc_summary.py $OUT > gpurun_out/pmcm_$TAG.txt
grep -A24 -E "k_paths<false|k_persist<8, true, false, true" gpurun_out/pmcm_$TAG.txt
