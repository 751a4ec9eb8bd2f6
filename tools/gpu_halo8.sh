#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
DVQ_HALO_WAVES=8 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "conv or blocks or dqvae or groupnorm" 2>&1 | tail -5 | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "vq" 2>&1 | tail -2
for w in 0 8; do DVQ_HALO_WAVES=$w timeout 300 python bench.py --steps 6 --warmup 1 --no-ae-only --no-cpu-baseline --no-vq-microbench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('halo waves=$w', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['ms_per_step'],v['TFLOPs']) for k,v in d['kernel_families'].items()})"; done
timeout 200 python bench.py --vq-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['vq_argmin']; print({k:(v['ms'], v['mfma_frac'], v['rerank_rows_full'], v['rerank_rows_candidates']) for k,v in d.items()})"
