#!/bin/bash
# generic same-box A/B of the headline step: bash tools/gpu/ab.sh "<env A>" "<env B>" ... (two alternating rounds, 10 steps each)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env $1 timeout 600 python bench.py --steps 10 --warmup 3 --no-ae-only --no-cpu-baseline --no-parity --no-fp32-mode --no-vq-microbench --no-extras 2>gpurun_out/ab_err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do for e in "$@"; do run "$e"; done; done | tee gpurun_out/ab_last.txt
