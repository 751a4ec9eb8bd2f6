#!/bin/bash
# the driver's default command, with its wall time
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; echo "exit $? wall $(( $(date +%s) - t0 )) s"
python - <<'P'
import json
for l in open("gpurun_out/default_bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["roofline"]["frac"], d["roofline"]["traffic_source"][-110:])
        print({k: (v.get("value"), v.get("skipped"), v.get("failed")) for k, v in d["extra_workloads"].items()})
        s = d["extra_workloads"]["sampling_p6c18"].get("by_batch_concurrent_lanes") or {}
        print({k: v.get("token_steps_per_sec") for k, v in s.items()})
P
