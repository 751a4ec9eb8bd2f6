#!/bin/bash
# measurement set of round 6 on ONE commit (TAG=r06_v1 HEAD_SHA=<sha> bash tools/gpu/final6.sh): parity suite, smoke, the driver's headline
# command, rocprofv3 kernel stats of the recorded single-stream step, FETCH / WRITE PMC passes (stamped with the commit and the
# kernel-source hash), per-shape tables (bf16 and fp32x3), conv table, GroupNorm probe, stage-2 STEADY-STATE step table (the first
# profiled step is dropped: its one-off zero fills are start-up, not step, work), GEMM probes against the library yardstick
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"; TAG=${TAG:-r06_v1}; export HEAD_SHA=${HEAD_SHA:-}
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
rm -rf gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE gpurun_out/prof_final
DVQ_SIDE_WGRAD=0 TOP=160 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_step_shapes.txt; echo "shapes exit $?"
DVQ_SHAPES_DTYPE=fp32x3 DVQ_SIDE_WGRAD=0 TOP=80 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_x3_step_shapes.txt; echo "x3 shapes exit $?"
timeout 300 python tools/conv_bench.py --no-check 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_conv_bench.txt; echo "conv_bench exit $?"
timeout 200 python tools/debug/gn_probe.py 2>/dev/null | grep "^N" > gpurun_out/${TAG}_gn_probe.txt
# stage-2 step, steady state: 1 profiled-away warm-up would still carry start-up fills, so profile 2 warm-up + 6 timed steps and divide the
# PER-DISPATCH trace by step: only dispatches after the first timed step's first attention forward count
rm -rf gpurun_out/prof_s2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s2 -o s2 -- python $R/bench_extra.py --workload stage2 --steps 6 --warmup 3 --no-cpu-baseline) > gpurun_out/prof_s2.log 2>&1
kt=$(find gpurun_out/prof_s2 -name "*kernel_trace.csv" | head -1)
python - "$kt" <<'P' > gpurun_out/${TAG}_stage2_step_table.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
att = [i for i, r in enumerate(rows) if "attn2_fwd_kernel<128" in r["Kernel_Name"] or "attn_fwd_kernel<128>" in r["Kernel_Name"]]
per_step = 24
nsteps = len(att) // per_step
keep_from = att[(nsteps - 5) * per_step]            # the last five steps: steady state
rows = rows[keep_from:]
acc, cnt = collections.Counter(), collections.Counter()
for r in rows:
    acc[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[r["Kernel_Name"]] += 1
tot = sum(acc.values())
print(f"steady state: last 5 of {nsteps} profiled steps; kernel time {tot / 5e6:.2f} ms per step (both streams summed), wall span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 5e6:.2f} ms per step")
for k, v in acc.most_common(40):
    print("%8.2f ms/step %7.1f calls/step  %s" % (v / 5e6, cnt[k] / 5.0, k[:110]))
P
rm -rf gpurun_out/prof_s2; echo "stage2 prof done"; head -3 gpurun_out/${TAG}_stage2_step_table.txt
{ echo "# causal attention, p6c18 geometry (B 32, T 648, 8 heads x 128, dropout 0.1): round-6 kernels (csrc/attention2.hip), drop mask / rehash / no dropout, then the first generation"
  for m in 1 0; do echo "## v2 MASK=$m"; MASK=$m timeout 200 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"; done
  echo "## v2 no dropout"; PDROP=0 timeout 200 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
  echo "## first generation (DVQ_ATTN_V2=0) MASK=0"; DVQ_ATTN_V2=0 MASK=0 timeout 200 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
  echo "# AttnBlock attention (B 64, T 1024, one head of 256): round-6 kernels, then the first generation"
  timeout 200 python tools/debug/attn_full_bench.py 2>&1 | grep "attn_full\|AttnBlock"
  echo "## first generation"; DVQ_ATTN_V2=0 timeout 200 python tools/debug/attn_full_bench.py 2>&1 | grep "attn_full"; } > gpurun_out/${TAG}_attention_probe.txt
{ for ms in 1 0; do echo "== DVQ_HALO_MFMA_STATS=$ms"; DVQ_HALO_MFMA_STATS=$ms timeout 120 python tools/debug/halo_stats_check.py 2>&1 | grep "^N"; done; } > gpurun_out/${TAG}_halo_stats_check.txt
timeout 300 python tools/gemm8p_probe.py 2>&1 | grep "^NT\|repeat" > gpurun_out/${TAG}_gemm_nt_probe.txt
{ timeout 200 python tools/probes/tn_probe.py 2>&1 | grep "TN\|DVQ"; DVQ_TN_WIDE_WGS=256 timeout 200 python tools/probes/tn_probe.py 2>&1 | grep "TN\|DVQ" | sed 's/^/[256 workgroups] /'; } > gpurun_out/${TAG}_gemm_tn_probe.txt
python - <<'P'
import json,os
tag=os.environ.get("TAG","r06_v1")
for l in open(f"gpurun_out/{tag}_bench.json"):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","host_issue_ms_per_step","step_mfma_frac")}, d["config"]["step_graph"]["timed_steps"], d["roofline"]["frac"], (d.get("ae_only") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
        print("fp32_mode", (d.get("fp32_mode") or {}).get("value"), "fp32x3_mode", (d.get("fp32x3_mode") or {}).get("value"))
        for k,v in (d.get("extra_workloads") or {}).items(): print(k, v.get("value"), v.get("unit"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"))
        print("vq", json.dumps({k:(v["ms"],v["rerank_rows_candidates"]) for k,v in (d.get("vq_argmin") or {}).items()}))
P
