#!/bin/bash
# dQ kernel with pieces removed (DVQ_ATTN_DBG bits: 1 refills, 2 element-wise, 4 second GEMM, 8 first GEMMs): kernel-trace durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R="$PWD"; export TMPDIR=/tmp
for d in ${VARIANTS:-0 1 2 4 8 3 7 15}; do
  rm -rf gpurun_out/attn_dbg; 
  DVQ_ATTN_DBG=$d REPS=3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/attn_dbg" -o a -- python tools/debug/attn_probe.py > gpurun_out/attn_dbg.log 2>&1
  f=$(find gpurun_out/attn_dbg -name "*kernel_stats.csv" | head -1)
  python - "$f" "$d" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_" in r["Name"] and "rowdot" not in r["Name"]:
        import re
        n=re.search(r"attn_\w+<[^>]*>", r["Name"]).group(0)
        print("dbg=%s %-36s %8.1f us" % (sys.argv[2], n, float(r["AverageNs"])/1e3))
P
done
