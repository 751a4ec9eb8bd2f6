#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 30 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; python - <<'P'
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_issue_ms_per_step","step_mfma_frac")}, d["config"]["step_graph"], d["roofline"]["frac"])
print(d["kernel_families"]); print(d["ae_only"]["value"]); print(json.dumps(d["vq_argmin"]))
P
tail -3 gpurun_out/bench.err
