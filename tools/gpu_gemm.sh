#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=short -x -k "gemm" 2>&1 | tail -4 | cut -c1-300
timeout 200 python tools/gemm_probe.py 2>&1 | grep "^NT\|^TN\|Error" | cut -c1-330
