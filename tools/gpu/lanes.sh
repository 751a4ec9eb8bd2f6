#!/bin/bash
# sampler throughput by number of concurrent lanes (Dualformer.sample_many): bs 8 and 50
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for l in 2 3 4 6; do
  DVQ_BENCH_LANES=$l timeout 600 python bench_extra.py --workload sampling 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('lanes $l', {k: (v.get('token_steps_per_sec'), v.get('roofline', {}).get('frac'), v.get('failed')) for k, v in d['by_batch_concurrent_lanes'].items()}, 'single', {k: v['token_steps_per_sec'] for k, v in d['by_batch'].items()})"
done | tee gpurun_out/r06_sampler_lanes.txt
