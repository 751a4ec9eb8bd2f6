#!/bin/bash
# round 5: fp32x3 forward / input gradient on the halo kernel (planes on the channel axis, fp32 output) -- same-box A/B of the fp32x3 step,
# per-shape table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="--dtype fp32x3 --no-graph --steps 6 --warmup 2 --no-extras --no-fp32-mode --no-parity --no-cpu-baseline --no-ae-only --no-vq-microbench"
for m in 1 0 1 0; do
  DVQ_X3_HALO=$m timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('x3_halo=$m', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done | tee gpurun_out/r5_x3_halo_ab.txt
echo
