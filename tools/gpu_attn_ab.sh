#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in gemm fused; do echo "DVQ_ATTNBLOCK_BWD=$m"; DVQ_ATTNBLOCK_BWD=$m bash tools/gpu_bench_quick.sh 2>&1 | grep -v "halo\|vq_arg" ; done
DVQ_ATTNBLOCK_BWD=gemm timeout 300 python -m pytest tests/test_gpu_attention.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -x -k "attnblock or blocks or dqvae" 2>&1 | tail -3 | cut -c1-200
