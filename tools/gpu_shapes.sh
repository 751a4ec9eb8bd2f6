#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
TOP=70 timeout 300 python tools/debug/step_shapes.py > gpurun_out/step_shapes_r2.txt 2>&1; echo "shapes exit $?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_attn" -o attn -- python "$R/tools/debug/attn_full_bench.py" > "$R/gpurun_out/prof_attn.log" 2>&1); echo "prof exit $?"
python - <<'P'
import csv,glob,re
for f in glob.glob('gpurun_out/prof_attn/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(re.sub(r'\(anonymous namespace\)::','',r['Name'])[:70], r['Calls'], float(r['AverageNs'])/1e3)
P
