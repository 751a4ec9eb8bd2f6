#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" MASTER_ADDR=127.0.0.1 timeout 300 python tests/dp_graph_check.py 29611 ${MODE:-graph} > gpurun_out/dp_$tag.log 2>&1; echo "== $tag exit $?"; grep "replayed item\|Memory access\|STAGE graph\|DP_GRAPH_OK\|Error\|assert" gpurun_out/dp_$tag.log | tail -4; }
run fine DVQ_GRAPH_DEBUG=1 DVQ_FORCE_DP=1 DVQ_DP_NOOP_COLLECTIVES=1 DVQ_DP_NO_HOOK=1 DVQ_GRAPH_BREAK_EVERY=40
MODE=both run real DVQ_FORCE_DP=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 25 gpurun_out/pytest_gpu.log
