#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stepgraph.py tests/test_gpu_model.py tests/test_gpu_lossnet.py tests/test_gpu_data.py -m gpu -q -p no:cacheprovider --tb=short -x --timeout 600 2>&1 | tail -3 | cut -c1-300
echo "graph (default)"; bash tools/gpu_bench_quick.sh 2>&1 | head -1
echo "eager (default side on)"; bash tools/gpu_bench_quick.sh --no-graph 2>&1 | head -1
