#!/bin/bash
# usage: tools/pmc3.sh <tag> "<bench args>"  — SQ / LDS / TA counters of the bench command in separate --pmc passes; per-kernel
# sums land in gpurun_out/pmc3_<tag>.txt (copy into profiles/ what should be judged).
TAG=$1; ARGS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc3_$TAG; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > $OUT/run$i.log 2>&1
  echo "pass $i exit $?"
done
python3 tools/pmc_summary.py $OUT > gpurun_out/pmc3_$TAG.txt
grep -A26 -E "k_paths<false|k_persist<8, true, false, true" gpurun_out/pmc3_$TAG.txt
