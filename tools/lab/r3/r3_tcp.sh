#!/bin/bash
# Is the vector-memory path (TA / TCP) what the path kernel waits for?  (i) a diagnostic build with two fewer table loads per pass
# (tools/variant.sh noinv rt_paths.hip -DRT_DIAG_NO_INV_LOADS: wrong frames, timing only) against the shipped library;
# (ii) TA / TCP / TD busy and stall counters of the headline frame, one or two per pass (larger sets are refused: tools/pmc_mem.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=raytrace_amd
{ echo "# headline: shipped / two table loads per pass removed / byte loads skipped when no lane needs one"; tools/abn.sh 3 $L/librt_amd.so $L/librt_amd_noinv.so $L/librt_amd_skipld.so
  echo "# small frames on k_paths: shipped / skipld"
  for size in "--width 1024 --height 1024 --spp 1 --depth 2" "--spp 1 --depth 4" "--spp 2 --depth 4" "--width 256 --height 256 --spp 1 --depth 2"; do
    echo "## $size"; BENCH_ARGS="$size --steps 50 --warmup 5 --kernel paths" tools/abn.sh 1 $L/librt_amd.so $L/librt_amd_skipld.so
  done; } 2>&1 | tee gpurun_out/r3_noinv_ab.txt
OUT=gpurun_out/pmc_tcp; rm -rf $OUT; mkdir -p $OUT
i=0
for P in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TD_TD_BUSY_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/run$i.log 2>&1
  echo "pass $i ($P) exit $?"
done
python3 tools/pmc_summary.py $OUT > gpurun_out/r3_pmc_tcp.txt 2>&1
grep -A22 -E "k_paths<false" gpurun_out/r3_pmc_tcp.txt | head -40
