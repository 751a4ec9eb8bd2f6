#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -x --timeout 600 -k "conv or blocks or halo or dqvae" 2>&1 | tail -3 | cut -c1-300; done
PROBE_REPS=10 timeout 300 python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
PROBE_REPS=20 PROBE_C=256 PROBE_H=64 timeout 300 python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
PROBE_REPS=20 PROBE_C=256 PROBE_H=32 timeout 300 python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
