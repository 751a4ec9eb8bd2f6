#!/bin/bash
# usage: gpu_ab_env.sh VAR val1 val2 ...   -> quick bench per value
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
V=$1; shift
for x in "$@"; do echo "$V=$x"; env $V=$x bash tools/gpu_bench_quick.sh 2>&1 | head -4; done
