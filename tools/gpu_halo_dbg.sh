#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for d in 0 1 2 3 4; do echo -n "DVQ_HALO_DBG=$d  "; DVQ_HALO_DBG=$d PROBE_REPS=10 timeout 200 python tools/conv_probe.py 2>/dev/null | tail -1; done
echo "C=256 H=64:"; for d in 0 3; do echo -n "DVQ_HALO_DBG=$d  "; DVQ_HALO_DBG=$d PROBE_C=256 PROBE_H=64 PROBE_REPS=20 timeout 200 python tools/conv_probe.py 2>/dev/null | tail -1; done
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_data.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "vq or train_py" 2>&1 | tail -4 | cut -c1-300
timeout 200 python bench.py --vq-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['vq_argmin']; print({k:(v['ms'], v['mfma_frac'], v['rerank_rows_candidates']) for k,v in d.items()})"
