#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
DVQ_HALO_DBG=5 PROBE_REPS=3 timeout 200 python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | grep "MHz" | sort | uniq -c | sort -rn | head -12
DVQ_HALO_DBG=5 PROBE_REPS=3 timeout 200 python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
