#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_featrouted.py -m gpu -q -p no:cacheprovider --tb=short -x -k "vq or quantiz or argmin or dqvae or config4" 2>&1 | tail -4 | cut -c1-250
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ae-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_families']['vq_argmin']); print({k:(v['ms'], v['rerank_rows_full'], v['rerank_rows_candidates']) for k,v in d['vq_argmin'].items()})"
