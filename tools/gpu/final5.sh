#!/bin/bash
# measurement set of round 5 on ONE commit (TAG=r05_v1 HEAD_SHA=<sha> bash tools/gpu/final5.sh): parity suite, smoke, the driver's headline
# command (fp32_mode / parity_bf16_vs_reference / extra_workloads blocks), rocprofv3 kernel stats of the recorded single-stream step,
# FETCH / WRITE PMC passes (stamped with the commit and the kernel-source hash), per-shape table (forward split by variant), conv
# table, halo-conv probes (probe library), GroupNorm probe, stage-2 step table, precision probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"; TAG=${TAG:-r05_v1}; export HEAD_SHA=${HEAD_SHA:-}
rm -f gpurun_out/test_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
cp gpurun_out/test_reports.jsonl gpurun_out/${TAG}_test_reports.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${TAG}_smoke.log | cut -c1-200
timeout 1500 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
rm -rf gpurun_out/prof_final
DVQ_SIDE_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_final" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --mode graph --no-cpu-baseline --no-ae-only --no-vq-microbench --no-extras --no-fp32-mode --no-parity > "$R/gpurun_out/prof_bench.log" 2>&1; echo "rocprof exit $?"
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcb_$c
  DVQ_SIDE_WGRAD=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcb_$c" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-ae-only --no-cpu-baseline --no-vq-microbench --no-extras --no-fp32-mode --no-parity > "$R/gpurun_out/pmcb_$c.log" 2>&1; echo "pmc $c exit $?"
done
ff=$(find gpurun_out/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summarise.py "$ff" "$fw" gpurun_out/${TAG}_bench_pmc.json | tail -4
DVQ_SIDE_WGRAD=0 TOP=160 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_step_shapes.txt; echo "shapes exit $?"
DVQ_SHAPES_DTYPE=fp32x3 DVQ_SIDE_WGRAD=0 TOP=80 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_x3_step_shapes.txt; echo "x3 shapes exit $?"
timeout 300 python tools/conv_bench.py --no-check 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_conv_bench.txt; echo "conv_bench exit $?"
if [ -f dynamicvectorquantization_amd/libdvq_hip_probes.so ]; then
{ echo "# 128->128 @256^2 B=64 halo conv (tools/debug/halo_data_probe.py, probe library): complete kernel / main loop only (DVQ_HALO_DBG=1) / epilogue only (=2)";
  for d in 0 1 2; do echo "## DVQ_HALO_DBG=$d"; DVQ_USE_PROBES_LIB=1 DVQ_HALO_DBG=$d timeout 200 python tools/debug/halo_data_probe.py 2>/dev/null | grep -A11 "epilogue options"; done; } > gpurun_out/${TAG}_halo_probes.txt 2>&1
fi
timeout 200 python tools/debug/gn_probe.py 2>/dev/null | grep "^N" > gpurun_out/${TAG}_gn_probe.txt
TOP=40 bash tools/gpu/r4_s2prof.sh > gpurun_out/${TAG}_stage2_step_table.txt 2>&1; cp gpurun_out/r04_stage2_kernel_stats.csv gpurun_out/${TAG}_stage2_kernel_stats.csv 2>/dev/null; echo "stage2 prof exit $?"
timeout 600 python tools/debug/r5_precision_probe.py 30 > gpurun_out/${TAG}_precision_probe.txt 2>&1; echo "probe exit $?"
python - <<'P'
import json,os
tag=os.environ.get("TAG","r05_v1")
for l in open(f"gpurun_out/{tag}_bench.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","host_issue_ms_per_step","step_mfma_frac")}, d["config"]["step_graph"]["timed_steps"], d["roofline"]["frac"], (d.get("ae_only") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
        print("fp32_mode", (d.get("fp32_mode") or {}).get("value"), ((d.get("fp32_mode") or {}).get("roofline") or {}).get("frac"))
        for k,v in (d.get("extra_workloads") or {}).items(): print(k, v.get("value"), v.get("unit"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), json.dumps(v.get("by_batch_concurrent_lanes"))[:400] if v.get("by_batch_concurrent_lanes") else "")
        print("vq", json.dumps({k:(v["ms"],v["rerank_rows_candidates"]) for k,v in (d.get("vq_argmin") or {}).items()}))
P
