#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_attention.py -m gpu -q -p no:cacheprovider --tb=short -x --timeout 600 -k "conv or blocks or halo or dqvae or gemm or attnblock" 2>&1 | tail -5 | cut -c1-300
bash tools/gpu_bench_quick.sh
