#!/bin/bash
# round 5, call C: full parity suite (deterministic mode, fixed tolerance tests), VQ micro-benchmark warm (second run counts),
# stage-2 split-count A/B, sampling with 2 / 3 lanes, quick headline with the decoder-only parity field
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_c}
rm -f gpurun_out/test_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 6 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
cp gpurun_out/test_reports.jsonl gpurun_out/${TAG}_test_reports.jsonl 2>/dev/null
for r in 1 2; do timeout 200 python bench.py --vq-only > gpurun_out/${TAG}_vq_only_$r.json 2>/dev/null; python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_vq_only_$r.json').read().strip().splitlines()[-1])['vq_argmin']
print('vq-only run $r', {k:(v['ms'],v['rerank_rows_candidates'],v['rerank_rows_wide']) for k,v in d.items()})"; done
for w in 256 128 64; do
  DVQ_TN_WIDE_WGS=$w timeout 400 python bench_extra.py --workload stage2 --steps 4 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_stage2_wgs$w.json
  python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_stage2_wgs$w.json').read())
print('stage2 DVQ_TN_WIDE_WGS=$w', d['value'], d['ms_per_step'], d.get('mfma_frac_est'))"
done
for l in 2 3; do
DVQ_BENCH_LANES=$l timeout 500 python bench_extra.py --workload sampling --no-cpu-baseline 2>gpurun_out/${TAG}_sampling.err | tail -1 > gpurun_out/${TAG}_sampling_l$l.json; echo "sampling exit $?"
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_sampling_l$l.json').read())
print({k:v['token_steps_per_sec'] for k,v in d['by_batch'].items()}, {k:(v.get('token_steps_per_sec'), v.get('failed')) for k,v in d.get('by_batch_concurrent_lanes',{}).items()})"
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-fp32-mode > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<'P'
import json,os
tag=os.environ.get("TAG","r05_c")
for l in open(f"gpurun_out/{tag}_bench.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], (d.get("ae_only") or {}).get("value"))
        print("parity_ref", json.dumps(d.get("parity_bf16_vs_reference"))[:1800])
        print("vq", json.dumps({k:(v["ms"],v["rerank_rows_candidates"]) for k,v in (d.get("vq_argmin") or {}).items()}))
P
