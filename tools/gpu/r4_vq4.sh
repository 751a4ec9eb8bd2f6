#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=short -rf -x --timeout 600 -k "vq" > gpurun_out/r4_vq_pytest.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/r4_vq_pytest.log | cut -c1-300
VARIANTS="0 1 29" bash tools/gpu/r4_vq2.sh
timeout 300 python bench.py --vq-only 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vq_argmin']
for k,v in d.items(): print(k, v['ms'], 'ms', v['GBps'], 'GB/s mfma', v['mfma_frac'], 'rerank', v['rerank_rows_full'], v['rerank_rows_candidates'], v.get('rerank_rows_wide'))
"
