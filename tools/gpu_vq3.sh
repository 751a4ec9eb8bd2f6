#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
for v in 0 1 2; do DVQ_VQ_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=line --timeout 600 -k "vq" 2>&1 | tail -1; done
for v in 0 1 2; do
  cd /tmp; DVQ_VQ_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vqv$v" -o vq -- python "$R/bench.py" --vq-only > "$R/gpurun_out/prof_vqv$v.log" 2>&1; cd "$R"
  f=$(find gpurun_out/prof_vqv$v -name "*kernel_stats.csv" | head -1); echo "== variant $v"; python - "$f" gpurun_out/prof_vqv$v.log <<'P'
import csv,sys,json
for x in csv.DictReader(open(sys.argv[1])):
    if 'argmin' in x['Name'] or 'rerank' in x['Name'] or 'zero' in x['Name']:
        print('  ', x['Name'][28:84].replace('(anonymous namespace)::',''), round(float(x['AverageNs'])/1000,1))
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])['vq_argmin']; print('  ', {k:(v['ms'],v['mfma_frac']) for k,v in d.items()})
except Exception as e: print(e)
P
done
