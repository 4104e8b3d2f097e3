#!/bin/bash
# usage: tools_pmc.sh "<bench args>" "<counter list>" ; prints per-kernel counter sums
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc
timeout -k 10 300 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmc -- python3 bench.py $1 --no-cpu-baseline > gpurun_out/pmc_run.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc | grep -A12 -E "k_mega<false|k_persist2?<8, true, false"
