#!/bin/bash
# HBM traffic of the bench's kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same bench.py command,
# counters only + --kernel-trace (no other trace domains).  MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE counts 64 B per 128-B
# request of wide streaming reads -> doubled in tools/pmc_summarise.py.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcb_$c" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-ae-only --no-cpu-baseline --no-vq-microbench > "$R/gpurun_out/pmcb_$c.log" 2>&1; echo "pmc $c exit $?"
  ls "$R/gpurun_out/pmcb_$c" | head -5
done
