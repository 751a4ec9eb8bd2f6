#!/bin/bash
# PMC passes on the conv micro-probe (GPU box).  Counters only, with --kernel-trace (no other trace domains).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
python tools/conv_probe.py 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -o p -- python "$R/tools/conv_probe.py" > "$R/gpurun_out/pmc_fetch.log" 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_hit" -o p -- python "$R/tools/conv_probe.py" > "$R/gpurun_out/pmc_hit.log" 2>&1; echo "pmc hit exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq" -o p -- python "$R/tools/conv_probe.py" > "$R/gpurun_out/pmc_sq.log" 2>&1; echo "pmc sq exit $?"
cd "$R"; find gpurun_out/pmc_* -name "*.csv" | head
