#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_data.py -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 -k "vq or data or transform or loader" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_sel.log | cut -c1-300
DVQ_VQ_2W=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --tb=line --timeout 600 -k "vq" 2>&1 | tail -2
for tag in base 2w v1; do
  cd /tmp; e=""; [ $tag = 2w ] && export DVQ_VQ_2W=1; [ $tag = v1 ] && export DVQ_VQ_V1=1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vq_$tag" -o vq -- python "$R/bench.py" --vq-only > "$R/gpurun_out/prof_vq_$tag.log" 2>&1; cd "$R"; unset DVQ_VQ_2W DVQ_VQ_V1
  f=$(find gpurun_out/prof_vq_$tag -name "*kernel_stats.csv" | head -1); echo "== $tag"; python - "$f" gpurun_out/prof_vq_$tag.log <<'P'
import csv,sys,json
for x in csv.DictReader(open(sys.argv[1])):
    if 'argmin' in x['Name'] or 'rerank' in x['Name'] or 'zero' in x['Name']:
        print('  ', x['Name'][28:80].replace('(anonymous namespace)::',''), round(float(x['AverageNs'])/1000,1))
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])['vq_argmin']; print('  ', {k:(v['ms'],v['mfma_frac']) for k,v in d.items()})
except Exception as e: print(e)
P
done
