#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lossnet.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 -k "vq or adaptive or dqvae_forward" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -n 12 gpurun_out/pytest_sel.log | cut -c1-400
for v in 0 1; do DVQ_VQ_V1=$v timeout 200 python bench.py --vq-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['vq_argmin']
print('V1=$v', {k:(v['ms'], v['mfma_frac']) for k,v in d.items()})"; done
cat gpurun_out/test_reports.jsonl | tail -3
