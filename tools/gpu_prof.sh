#!/bin/bash
# rocprofv3 kernel stats of a short bench run (GPU box)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$PWD"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$R/gpurun_out/prof_bench.log" 2>&1
echo "rocprof exit $?"; cd "$R"; tail -2 gpurun_out/prof_bench.log | cut -c1-600
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo "$f"; head -22 "$f" | cut -c1-220
