#!/bin/bash
# quick kernel iteration loop on the GPU box: selected parity tests + conv micro-probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 300 -k "${K:-conv or gemm or blocks or dqvae}" > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?"; tail -n ${TAILN:-8} gpurun_out/pytest_quick.log
timeout 300 python tools/conv_probe.py 2>&1 | tail -1
