#!/bin/bash
# rocprofv3 kernel stats of the stage-2 (DQ-Transformer p6c18) train step -> gpurun_out/r04_stage2_kernel_stats.csv + a per-step table
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R="$PWD"; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_s2
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s2 -o s2 -- python $R/bench_extra.py --workload stage2 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_s2.log 2>&1
cd $R; f=$(find gpurun_out/prof_s2 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_stage2_kernel_stats.csv
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
att=[r for r in rows if "attn_fwd_kernel<128>" in r["Name"]]
steps=int(att[0]["Calls"])/24 if att else 7
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("steps", steps, "total ms per step", tot/1e6/steps)
for r in rows[:int(__import__("os").environ.get("TOP","32"))]:
    print("%8.2f ms/step %7.1f calls/step  %s" % (float(r["TotalDurationNs"])/1e6/steps, int(r["Calls"])/steps, r["Name"][:100]))
P
