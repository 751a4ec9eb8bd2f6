#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
for d in 0 1 2 3 4 7; do
  cd /tmp; DVQ_VQ_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vq$d" -o vq -- python "$R/bench.py" --vq-only > "$R/gpurun_out/prof_vq$d.log" 2>&1; cd "$R"
  f=$(find gpurun_out/prof_vq$d -name "*kernel_stats.csv" 2>/dev/null | head -1); echo "dbg=$d"; [ -n "$f" ] && grep "pipe_kernel" "$f" | awk -F'","' '{print substr($1,1,90), $4}' || tail -3 gpurun_out/prof_vq$d.log
done
