#!/bin/bash
# same-box A/B of the workgroup targets of the side-stream weight-gradient kernels (headline step, alternating rounds)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-ae-only --no-cpu-baseline --no-parity --no-fp32-mode --no-vq-microbench --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  run DVQ_WGRAD_WGS=256
  run DVQ_WGRAD_WGS=128
  run DVQ_WGRAD_WGS=128 DVQ_TN_PATCH_WGS=384
  run DVQ_WGRAD_WGS=128 DVQ_TN_PATCH_WGS=256
  run DVQ_WGRAD_WGS=128 DVQ_TN_PATCH_WGS=128
done | tee gpurun_out/r06_wgrad_wgs_ab.txt
