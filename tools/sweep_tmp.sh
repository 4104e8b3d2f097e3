mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
bash tools/profile.sh > gpurun_out/profile.log 2>&1; grep -n "exit" gpurun_out/profile.log | head -4
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
D="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
for k in persistent persistent2; do echo "== $k"; bash tools/pmc.sh "--kernel $k --steps 2 --warmup 1" "$C" | grep -v "^void"; bash tools/pmc.sh "--kernel $k --steps 2 --warmup 1" "$D" | grep -v "^void"; done
bash tools/bench_configs.sh > /dev/null 2>&1; cut -c1-150 gpurun_out/bench_configs.jsonl | grep ms_per_step | sed 's/.*"value"/value/' 
