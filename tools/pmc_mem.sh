#!/bin/bash
# usage: tools/pmc_mem.sh <tag> "<bench args>" — vector-memory pipeline counters (TA / TCP / UTCL1) of the bench command
TAG=$1; ARGS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcm_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
# NB (diagnosed in round 3 from the round-2 logs, gpurun_out/pmcm_paths/run{1,2,5}.log): the passes that asked for four TA_* /
# TD_* / TCP_GATE_EN* / TCP_PENDING_STALL counters at once did not hang the GPU — rocprofv3 refused the set before any kernel ran
# ("rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect": more
# counters of one block than it has registers), aborted (signal 6) inside its own tool library, and its signal handler then sat
# until the 240 s limit killed the process.  No dispatch had been made, nothing was left on the device.  Such counters have to
# be asked for in smaller sets (one or two per block and pass); the two sets below fit and return.
for P in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > $OUT/run$i.log 2>&1
  echo "pass $i exit $?"
done
python3 tools/pmc_summary.py $OUT > gpurun_out/pmcm_$TAG.txt
grep -A24 -E "k_paths<false|k_persist<8, true, false, true" gpurun_out/pmcm_$TAG.txt
