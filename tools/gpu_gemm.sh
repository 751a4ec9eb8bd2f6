#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -x -k "gemm or conv1x1 or blocks or dqvae" 2>&1 | tail -4 | cut -c1-300
for w in 0 1; do echo "DVQ_NO_WORKSPACE=$w"; DVQ_NO_WORKSPACE=$w timeout 200 python tools/gemm_probe.py 2>&1 | grep "^TN" | cut -c1-200; done
bash tools/gpu_bench_quick.sh 2>&1 | head -8
