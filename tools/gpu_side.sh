#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in auto graph eager; do echo "--mode $m"; timeout 400 python bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline --no-vq-microbench --no-ae-only 2>gpurun_out/bench_mode.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['step_graph'], d['roofline']['frac'])"; done
tail -3 gpurun_out/bench_mode.err | cut -c1-200
