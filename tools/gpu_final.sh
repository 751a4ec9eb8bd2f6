#!/bin/bash
# round-2 measurement set: parity suite, smoke, headline bench, rocprofv3 kernel stats, PMC traffic passes, secondary workloads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"; TAG=${TAG:-r02_v1}
rm -f gpurun_out/test_reports.jsonl
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cd /tmp; DVQ_SIDE_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_final" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --mode graph --no-cpu-baseline --no-ae-only --no-vq-microbench > "$R/gpurun_out/prof_bench.log" 2>&1; echo "rocprof exit $?"; cd "$R"
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  DVQ_SIDE_WGRAD=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcb_$c" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-ae-only --no-cpu-baseline --no-vq-microbench > "$R/gpurun_out/pmcb_$c.log" 2>&1; echo "pmc $c exit $?"
done
cd "$R"
ff=$(find gpurun_out/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summarise.py "$ff" "$fw" gpurun_out/${TAG}_bench_pmc.json | tail -4
: > gpurun_out/${TAG}_extra_workloads.jsonl
timeout 400 python bench_extra.py --workload triple --steps 6 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${TAG}_extra_workloads.jsonl
timeout 400 python bench_extra.py --workload triple --codebook 8192 --steps 6 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${TAG}_extra_workloads.jsonl
timeout 500 python bench_extra.py --workload stage2 --steps 4 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/${TAG}_extra_workloads.jsonl
timeout 400 python bench_extra.py --workload sampling 2>/dev/null | tail -1 >> gpurun_out/${TAG}_extra_workloads.jsonl
python - <<'P'
import json,os
tag=os.environ.get("TAG","r02_v1")
d=json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_issue_ms_per_step","step_mfma_frac")}, d["config"]["step_graph"], d["roofline"]["frac"], d["ae_only"]["value"], d["cpu_baseline"]["value"])
for l in open(f"gpurun_out/{tag}_extra_workloads.jsonl"):
    try:
        e=json.loads(l); print(e["workload"], e["value"], e["unit"], e.get("ms_per_step"), (e.get("roofline") or {}).get("kernel"), (e.get("roofline") or {}).get("frac"), e.get("config",{}).get("codebook"))
    except Exception as ex: print("bad line", ex)
P
