#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_stepgraph.py -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 600 -k "full_attention or attnblock or blocks or dqvae or step_graph_matches" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -n 12 gpurun_out/pytest_sel.log | cut -c1-300
for f in 0 1; do DVQ_NO_FUSED_ATTNBLOCK=$f timeout 300 python bench.py --steps 6 --warmup 1 --no-ae-only --no-cpu-baseline --no-vq-microbench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_fused_attnblock=$f', d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], {k:(v['launches'], v['ms_per_step'],v['TFLOPs']) for k,v in d['kernel_families'].items() if 'gemm' in k or 'attn' in k})"; done
