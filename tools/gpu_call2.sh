#!/bin/bash
# round-2 call: step-graph tests, DP check by stages, bench with / without step graphs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stepgraph.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 -k "stepgraph or step_graph or vq or sample_rows" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -n 30 gpurun_out/pytest_sel.log
for mode in eager graph; do
  DVQ_FORCE_DP=1 MASTER_ADDR=127.0.0.1 timeout 300 python tests/dp_graph_check.py 29611 $mode > gpurun_out/dp_$mode.log 2>&1; echo "dp $mode exit $?"; grep -v "amdgpu.ids\|hostname of the client" gpurun_out/dp_$mode.log | tail -8
done
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -c 6000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
timeout 600 python bench.py --steps 5 --warmup 1 --no-graph --no-ae-only --no-cpu-baseline --no-vq-microbench > gpurun_out/bench_nograph.log 2> gpurun_out/bench_nograph.err; echo "bench nograph exit $?"; python - <<'P'
import json
for f in ("gpurun_out/bench.log","gpurun_out/bench_nograph.log"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"].get("step_graph"), d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
P
