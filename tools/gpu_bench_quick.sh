#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-vq-microbench "$@" 2>gpurun_out/bench_quick.err | tail -1 > gpurun_out/bench_quick.json
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print(d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step'), d['roofline']['frac'], d['roofline']['achieved'])
for k,v in sorted(d['kernel_families'].items(), key=lambda kv:-kv[1]['ms_per_step']): print(' ', k, v)
P
