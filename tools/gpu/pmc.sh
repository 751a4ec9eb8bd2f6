#!/bin/bash
# rocprofv3 PMC pass of a command: usage pmc.sh <tag> "<counters>" <cmd...>; per-kernel sums land in gpurun_out/<tag>_pmc.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
tag=$1; ctr=$2; shift 2
# the command runs with the repo as working directory; only rocprofv3's own scratch lives under /tmp
rm -rf "$R/gpurun_out/pmc_$tag"
timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$tag" -o p -- "$@" > "$R/gpurun_out/pmc_$tag.log" 2>&1; echo "rocprof exit $?"
cd "$R"
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" <<'P' > gpurun_out/${tag}_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in d.items()})
P
cat gpurun_out/${tag}_pmc.txt | head -20
