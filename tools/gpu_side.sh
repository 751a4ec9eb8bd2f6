#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for i in 1 2 3; do timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vq-microbench 2>gpurun_out/bench_mode.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['step_graph']['timed_steps'], d['config']['step_graph']['calibration'], d['ae_only']['value'])"; done
