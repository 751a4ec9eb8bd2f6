#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_v3}
rm -f gpurun_out/test_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
cp gpurun_out/test_reports.jsonl gpurun_out/${TAG}_test_reports.jsonl 2>/dev/null
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_lossnet.py tests/test_gpu_stage2.py -m gpu -q -p no:cacheprovider --tb=short -k "lpips_lin_dropout or sample_many or dropout_backward" 2>&1 | tail -1; done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
