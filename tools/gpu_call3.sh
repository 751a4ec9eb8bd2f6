#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
DVQ_GRAPH_DEBUG=1 DVQ_FORCE_DP=1 MASTER_ADDR=127.0.0.1 timeout 300 python tests/dp_graph_check.py 29611 graph > gpurun_out/dp_graph.log 2>&1; echo "dp graph exit $?"; grep -v "amdgpu.ids\|hostname of the client" gpurun_out/dp_graph.log | tail -40
timeout 600 python tools/debug/graph_aa.py 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 25 gpurun_out/pytest_gpu.log
