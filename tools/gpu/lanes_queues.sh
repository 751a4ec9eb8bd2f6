#!/bin/bash
# sampler throughput by hardware-queue count (GPU_MAX_HW_QUEUES) x lanes -> gpurun_out/r06_sampler_hwq.txt (profiles/r06_sampler_lanes.txt, table 3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for q in 4 8 16; do for l in 2 4 6; do
  GPU_MAX_HW_QUEUES=$q DVQ_BENCH_LANES=$l timeout 600 python bench_extra.py --workload sampling 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('hwq $q lanes $l', {k: (v.get('token_steps_per_sec'), v.get('failed')) for k, v in d['by_batch_concurrent_lanes'].items()}, 'single', {k: v['token_steps_per_sec'] for k, v in d['by_batch'].items()})"
done; done | tee gpurun_out/r06_sampler_hwq.txt
