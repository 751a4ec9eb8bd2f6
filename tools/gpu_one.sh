#!/bin/bash
# usage: gpu_one.sh <python script and args...>   (stdout/stderr tail comes back through gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout ${LIMIT:-400} "$@" 2>&1 | grep -v "amdgpu.ids" | tail -n ${TAIL:-60} | cut -c1-400
