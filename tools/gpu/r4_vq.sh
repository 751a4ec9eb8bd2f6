#!/bin/bash
# round-4 VQ iteration: new tests, VQ micro-benchmark A/B (default = 64-row bf16 kernel; DVQ_VQ_VARIANT=3 = the 8-wave kernel of round 3;
# DVQ_VQ_DBG timing splits), kernel-level durations by rocprofv3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
rm -f gpurun_out/test_reports.jsonl
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stepgraph.py tests/test_gpu_stage2.py -m gpu -q -p no:cacheprovider --tb=short -rf -x --timeout 600 -k "vq or entropy or permuter or stepgraph or step_graph or eager_forwards or sample_rows" > gpurun_out/r4_vq_pytest.log 2>&1; echo "pytest exit $?"; tail -n 5 gpurun_out/r4_vq_pytest.log | cut -c1-300
for v in "" "DVQ_VQ_VARIANT=3" "DVQ_VQ_DBG=1" "DVQ_VQ_DBG=2" "DVQ_VQ_DBG=4"; do
  echo "== vq-only $v"
  env $v timeout 300 python bench.py --vq-only 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vq_argmin']
for k,v in d.items(): print(k, v['ms'], 'ms', v['GBps'], 'GB/s mfma', v['mfma_frac'], 'rerank', v['rerank_rows_full'], v['rerank_rows_candidates'], v.get('rerank_rows_wide'))
"
done
rm -rf gpurun_out/prof_vq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vq" -o vq -- python "$R/bench.py" --vq-only > gpurun_out/prof_vq.log 2>&1; echo "rocprof exit $?"
f=$(find gpurun_out/prof_vq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4_vq_kernel_stats.csv && head -8 "$f" | cut -c1-200
