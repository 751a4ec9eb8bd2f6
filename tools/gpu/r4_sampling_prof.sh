#!/bin/bash
# rocprofv3 kernel stats of the K/V-cached sampler at bs 8: kernel time per token step by kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R="$PWD"; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_smp /tmp/prof_smp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smp -o s -- python $R/tools/debug/sampling_profile.py > $R/gpurun_out/prof_smp.log 2>&1
cd $R; grep "token steps" gpurun_out/prof_smp.log
f=$(find /tmp/prof_smp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
dec=[r for r in rows if "decode_stack_kernel" in r["Name"]]
steps=int(dec[0]["Calls"])/2 if dec else 1
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("token steps (3 runs)", steps, "kernel us per token step", tot/1e3/steps)
for r in rows[:22]:
    print("%8.1f us/step %6.2f calls/step avg %7.1f us  %s" % (float(r["TotalDurationNs"])/1e3/steps, int(r["Calls"])/steps, float(r["AverageNs"])/1e3, r["Name"][:90]))
P
