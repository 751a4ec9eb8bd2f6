#!/bin/bash
# kernel trace of a 4-lane sampling run: device busy fraction, overlap between queues
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
rm -rf gpurun_out/prof_ln
(cd /tmp && DVQ_BENCH_LANES=4 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ln -o ln -- python $R/bench_extra.py --workload sampling --bs 8) > gpurun_out/prof_ln.log 2>&1
kt=$(find gpurun_out/prof_ln -name "*kernel_trace.csv" | head -1)
python - "$kt" <<'P'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last 20 % of the run = the 4-lane timed phase
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - (t1 - t0) * 0.12
win = [r for r in rows if r[0] >= lo]
qs = collections.Counter(r[2] for r in win)
span = max(r[1] for r in win) - win[0][0]
ev = []
for s, e, q in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, win[0][0], collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last; last = t; depth += d
print("window ms", span / 1e6, "launches", len(win), "queues", dict(qs))
print("time by number of kernels running at once:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
print("mean kernel us", sum(e - s for s, e, q in win) / len(win) / 1e3, "kernel-time sum / span", sum(e - s for s, e, q in win) / span)
P
rm -rf gpurun_out/prof_ln
