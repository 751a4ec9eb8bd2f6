#!/bin/bash
# full GPU suite + a short headline bench (no extras / cpu baseline)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
rm -f gpurun_out/test_reports.jsonl
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/r4_bench_quick.json 2> gpurun_out/bench.err; echo "bench exit $?"; tail -3 gpurun_out/bench.err | cut -c1-300
python - <<'P'
import json
for l in open("gpurun_out/r4_bench_quick.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","host_issue_ms_per_step","step_mfma_frac")}, d["config"]["step_graph"])
        print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "ae_only", (d.get("ae_only") or {}).get("value"))
        print("parity_bf16", d.get("parity_bf16"))
        for k,v in (d.get("vq_argmin") or {}).items(): print("vq", k, v["ms"], v["GBps"], v["mfma_frac"], v["rerank_rows_full"], v["rerank_rows_candidates"], v.get("rerank_rows_wide"))
        for k,v in sorted(d["kernel_families"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:14]: print("  ", k, v)
P
