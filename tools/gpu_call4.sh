#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" DVQ_GRAPH_DEBUG=1 DVQ_FORCE_DP=1 MASTER_ADDR=127.0.0.1 timeout 200 python tests/dp_graph_check.py 29611 graph > gpurun_out/dp_$tag.log 2>&1; echo "== $tag exit $?"; grep -v "amdgpu.ids\|hostname of the client\|UserWarning\|get_obj_from_str" gpurun_out/dp_$tag.log | grep -v "begin segment\|eager item" | tail -4; }
run noop_nohook DVQ_DP_NOOP_COLLECTIVES=1 DVQ_DP_NO_HOOK=1
run noop DVQ_DP_NOOP_COLLECTIVES=1
run nohook DVQ_DP_NO_HOOK=1
run seppools DVQ_GRAPH_SEPARATE_POOLS=1
run seppools_noop DVQ_GRAPH_SEPARATE_POOLS=1 DVQ_DP_NOOP_COLLECTIVES=1
