#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -x --timeout 600 -k "conv or blocks or halo or dqvae or gemm" 2>&1 | tail -3 | cut -c1-300
bash tools/gpu_bench_quick.sh 2>&1 | head -5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_q" -o q -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-ae-only --no-vq-microbench > /dev/null 2>&1)
python - <<'P'
import csv,glob,re
for f in glob.glob('gpurun_out/prof_q/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'reduce' in r['Name'] or 'finalize' in r['Name']: print(re.sub(r'\(anonymous namespace\)::','',r['Name'])[:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
P
