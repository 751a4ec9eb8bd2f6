#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 0 1; do
  HIP_FORCE_DEV_KERNARG=$k timeout 600 python bench_extra.py --workload sampling 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('HIP_FORCE_DEV_KERNARG=$k', {k: (v.get('token_steps_per_sec'), v.get('failed')) for k, v in d['by_batch_concurrent_lanes'].items()}, 'single', {k: v['token_steps_per_sec'] for k, v in d['by_batch'].items()})"
done
