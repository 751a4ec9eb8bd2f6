#!/bin/bash
# rocprofv3 kernel stats of the K/V-cached sampler at bs 8: kernel time per token step by kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R="$PWD"; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_smp /tmp/prof_smp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smp -o s -- python $R/tools/debug/sampling_profile.py > $R/gpurun_out/prof_smp.log 2>&1
cd $R; grep "token steps" gpurun_out/prof_smp.log
f=$(find /tmp/prof_smp -name "*kernel_stats.csv" | head -1)
python - "$f" gpurun_out/prof_smp.log <<'P'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
m=re.search(r"(\d+) token steps", open(sys.argv[2]).read())
steps=int(m.group(1))*3 if m else 1            # sampling_profile.py runs the sampler three times (warm-up, timed, cProfile)
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("token steps (3 runs)", steps, "kernel us per token step", round(tot/1e3/steps,1))
for r in rows[:22]:
    print("%8.1f us/step %6.2f calls/step avg %7.1f us  %s" % (float(r["TotalDurationNs"])/1e3/steps, int(r["Calls"])/steps, float(r["AverageNs"])/1e3, r["Name"][:90]))
P
