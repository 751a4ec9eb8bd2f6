#!/bin/bash
# launch-list replay of the recorded step: tests, then A/B against hipGraphLaunch and the eager step (same box, same process flags)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stepgraph.py -x -q 2>&1 | tail -15 > gpurun_out/r4_cmdlist_tests.txt
cat gpurun_out/r4_cmdlist_tests.txt
B="--steps 10 --warmup 2 --no-cpu-baseline --no-ae-only --no-vq-microbench --no-extras --no-parity"
: > gpurun_out/r4_cmdlist_ab.jsonl
run() {   # name, env..., -- args
    local name=$1; shift
    local out
    out=$(env "$@" timeout 600 python bench.py $B $MODE 2>gpurun_out/r4_cmdlist_$name.err | tail -1)
    python - "$name" "$out" "$*" "$MODE" >> gpurun_out/r4_cmdlist_ab.jsonl <<'PY'
import json, sys
name, line, env, mode = sys.argv[1:5]
try:
    j = json.loads(line)
    print(json.dumps({"run": name, "value": j["value"], "ms_per_step": j["ms_per_step"], "step_graph": j["config"].get("step_graph"),
                      "host_issue_ms_per_step": j.get("host_issue_ms_per_step"), "env": env, "args": mode}))
except Exception as e:
    print(json.dumps({"run": name, "error": str(e), "tail": line[-400:]}))
PY
    tail -3 gpurun_out/r4_cmdlist_$name.err
}
MODE="--mode graph" run list_replay DVQ_STEP_REPLAY=list
if [ -z "$QUICK" ]; then
MODE="--mode graph" run list_replay_module_launch DVQ_STEP_REPLAY=list DVQ_CMDLIST_LAUNCH=module
MODE="--mode graph" run hipgraph_replay DVQ_STEP_REPLAY=graph
MODE="--mode eager" run eager DVQ_STEP_REPLAY=list
MODE="--mode graph" run one_rank_dp_list DVQ_STEP_REPLAY=list DVQ_FORCE_DP=1
MODE="--mode graph" run one_rank_dp_hipgraph DVQ_STEP_REPLAY=graph DVQ_FORCE_DP=1
fi
cat gpurun_out/r4_cmdlist_ab.jsonl
