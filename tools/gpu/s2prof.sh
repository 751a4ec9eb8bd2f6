cd "${GRAFT_REPO_ROOT:-/root/repo}"; R="$PWD"; TAG=r06_v2b; export TMPDIR=/tmp
rm -rf gpurun_out/prof_s2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s2 -o s2 -- python $R/bench_extra.py --workload stage2 --steps 6 --warmup 3 --no-cpu-baseline) > gpurun_out/prof_s2.log 2>&1
kt=$(find gpurun_out/prof_s2 -name "*kernel_trace.csv" | head -1)
python - "$kt" <<'P' > gpurun_out/${TAG}_stage2_step_table.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
att = [i for i, r in enumerate(rows) if "attn2_fwd_kernel<128" in r["Kernel_Name"] or "attn_fwd_kernel<128>" in r["Kernel_Name"]]
per_step = 24
nsteps = len(att) // per_step
keep_from = att[(nsteps - 5) * per_step]
rows = rows[keep_from:]
acc, cnt = collections.Counter(), collections.Counter()
for r in rows:
    acc[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[r["Kernel_Name"]] += 1
tot = sum(acc.values())
print(f"steady state: last 5 of {nsteps} profiled steps; kernel time {tot / 5e6:.2f} ms per step (both streams summed), wall span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 5e6:.2f} ms per step")
for k, v in acc.most_common(40):
    print("%8.2f ms/step %7.1f calls/step  %s" % (v / 5e6, cnt[k] / 5.0, k[:110]))
P
rm -rf gpurun_out/prof_s2; head -32 gpurun_out/${TAG}_stage2_step_table.txt
