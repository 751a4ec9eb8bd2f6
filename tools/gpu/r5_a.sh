#!/bin/bash
# round 5, call A: parity suite + smoke on the new boundary code, the headline with the fp32_mode / parity_vs_reference blocks, the
# precision / VQ-operand probe, per-shape table with the forward split by variant
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_a}
rm -f gpurun_out/test_reports.jsonl
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
cp gpurun_out/test_reports.jsonl gpurun_out/${TAG}_test_reports.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${TAG}_smoke.log | cut -c1-300
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"; tail -c 600 gpurun_out/${TAG}_bench.err
timeout 900 python tools/debug/r5_precision_probe.py 30 > gpurun_out/${TAG}_precision_probe.txt 2>&1; echo "probe exit $?"; tail -n 12 gpurun_out/${TAG}_precision_probe.txt | cut -c1-400
DVQ_SIDE_WGRAD=0 TOP=160 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_step_shapes.txt; echo "shapes exit $?"
python - <<'P'
import json,os
tag=os.environ.get("TAG","r05_a")
for l in open(f"gpurun_out/{tag}_bench.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], (d.get("ae_only") or {}).get("value"))
        print("fp32_mode", json.dumps(d.get("fp32_mode"))[:900])
        print("parity_ref", json.dumps(d.get("parity_bf16_vs_reference"))[:1500])
        print("parity_bf16", json.dumps(d.get("parity_bf16"))[:600])
        print("vq", json.dumps({k:(v["ms"],v["rerank_rows_candidates"]) for k,v in (d.get("vq_argmin") or {}).items()}))
P
