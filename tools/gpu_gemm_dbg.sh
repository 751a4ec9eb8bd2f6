#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for d in 0 1 4 8 12; do echo -n "DVQ_GEMM_DBG=$d  "; DVQ_GEMM_DBG=$d timeout 120 python tools/debug/gemm_dbg_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "main loop" | tr '\n' ' '; echo; done
