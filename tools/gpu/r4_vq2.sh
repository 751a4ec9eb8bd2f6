#!/bin/bash
# VQ bf16 main-kernel timing splits (DVQ_VQ_DBG bits: 1 no bookkeeping, 2 no MFMA, 4 no DMA, 8 no fragment reads, 16 no stage barrier)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
for v in ${VARIANTS:-0 1 8 9 13 16 29 2}; do
  rm -rf gpurun_out/prof_vq
  DVQ_VQ_DBG=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vq" -o vq -- python "$R/bench.py" --vq-only > gpurun_out/prof_vq.log 2>&1
  f=$(find gpurun_out/prof_vq -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[f"DVQ_VQ_DBG={sys.argv[2]}"]
for r in rows:
    n=r["Name"]
    if "rb2" in n or "rerank" in n or "pipe_kernel" in n:
        short="rb2" if "rb2" in n else ("rerank_bf16" if "rerank" in n and "unsigned short" in n else ("rerank_f32" if "rerank" in n else "pipe8_f32"))
        out.append(f"{short}: {float(r['AverageNs'])/1e3:.1f} us x{r['Calls']}")
print("  ".join(out))
P
done
