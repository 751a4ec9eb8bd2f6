#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" DVQ_GRAPH_DEBUG=1 MASTER_ADDR=127.0.0.1 timeout 300 python tests/dp_graph_check.py 29611 graph > gpurun_out/dp_$tag.log 2>&1; echo "== $tag exit $?"; grep "replayed item\|Memory access\|STAGE graph" gpurun_out/dp_$tag.log | tail -3; }
# (1) exchange points without RCCL calls, no in-backward hook: the plain multi-segment case, cut every 40 library calls
run fine DVQ_FORCE_DP=1 DVQ_DP_NOOP_COLLECTIVES=1 DVQ_DP_NO_HOOK=1 DVQ_GRAPH_BREAK_EVERY=40
# (2) NO data-parallel at all (one rank, no exchange points) but the step cut every 200 calls: is segmentation alone enough?
DVQ_GRAPH_DEBUG=1 DVQ_GRAPH_BREAK_EVERY=200 timeout 300 python - > gpurun_out/dp_nodp.log 2>&1 <<'P'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_stepgraph as T
m, tr, l = T._run(torch.device("cuda:0"), True, 6, "full")
print("NODP OK", l[-1].tolist(), tr._graph["sg"].n_segments(), flush=True)
P
echo "== nodp exit $?"; grep "replayed item\|Memory access\|NODP" gpurun_out/dp_nodp.log | tail -3
