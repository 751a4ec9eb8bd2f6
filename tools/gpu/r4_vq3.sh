#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
for ord in ${ORDS:-0 2}; do export DVQ_VQ_ORD=$ord; echo "== ORD $ord"; VARIANTS="${VARIANTS:-0 29}" bash tools/gpu/r4_vq2.sh; done
