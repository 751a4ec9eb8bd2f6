#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_f}
timeout 900 python tools/debug/r5_loss_tracking.py 16 > gpurun_out/${TAG}_loss_tracking.txt 2>&1; echo "loss tracking exit $?"; grep -v "Warn\|return get_obj" gpurun_out/${TAG}_loss_tracking.txt | tail -12 | cut -c1-600
