#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_i}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -k "fp32 or golden or deterministic" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -n 12 gpurun_out/${TAG}_pytest.log | cut -c1-400
timeout 900 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-ae-only --no-vq-microbench > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<'P'
import json,os
tag=os.environ.get("TAG","r05_i")
for l in open(f"gpurun_out/{tag}_bench.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'])
        for k in ("fp32_mode","fp32x3_mode"):
            v=d.get(k) or {}
            print(k, v.get('value'), v.get('ms_per_step'), v.get('roofline'), v.get('failed'))
            print('   ', {kk:(vv['launches'],vv['ms_per_step'],vv['TFLOPs']) for kk,vv in (v.get('kernel_families') or {}).items()})
        pr=d.get('parity_bf16_vs_reference') or {}
        for k in ("bf16","fp32","fp32x3"): print(k, json.dumps(pr.get(k))[:700])
P
