#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do for q in 2 3 4; do
GPU_MAX_HW_QUEUES=$q timeout 600 python bench_extra.py --workload triple --codebook 8192 --steps 4 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('triple hwq $q', d['value'], d.get('ms_per_step'))"
done; done
bash tools/gpu/ab.sh GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=3 GPU_MAX_HW_QUEUES=4
