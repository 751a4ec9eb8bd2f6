#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"; TAG=${TAG:-r02_v7}
rm -rf gpurun_out/prof_final
cd /tmp; DVQ_SIDE_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_final" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --mode graph --no-cpu-baseline --no-ae-only --no-vq-microbench > "$R/gpurun_out/prof_bench.log" 2>&1; echo "rocprof exit $?"; cd "$R"
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv && echo copied
