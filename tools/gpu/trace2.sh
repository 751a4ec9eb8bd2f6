#!/bin/bash
# kernel trace of the two-stream headline step (launch-list replay): per-dispatch start / end with queue ids -> gpurun_out/two_stream_trace.csv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
rm -rf gpurun_out/prof_t2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_t2 -o t2 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ae-only --no-vq-microbench --no-extras --no-fp32-mode --no-parity) > gpurun_out/prof_t2.log 2>&1
kt=$(find gpurun_out/prof_t2 -name "*kernel_trace.csv" | head -1)
python - "$kt" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~2 steps worth: find the adam kernels as step delimiters
ad = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r["Kernel_Name"]]
print("adam launches", len(ad))
# a step has 2 adam launches; take the window between the 4th-last and 2nd-last adam (one full step)
lo, hi = ad[-5] + 1, ad[-3] + 1
win = rows[lo:hi]
import gzip
with open("gpurun_out/two_stream_trace.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["start_ns", "end_ns", "queue", "kernel"])
    t0 = int(win[0]["Start_Timestamp"])
    for r in win:
        w.writerow([int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]])
print("window launches", len(win), "span ms", (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6)
P
rm -rf gpurun_out/prof_t2
