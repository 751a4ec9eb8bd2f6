#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_g}
timeout 600 python tools/debug/r5_loss_tracking.py 16 B > gpurun_out/${TAG}_loss_tracking_B.txt 2>&1; echo "loss tracking B exit $?"; grep -v "Warn\|return get_obj" gpurun_out/${TAG}_loss_tracking_B.txt | tail -4 | cut -c1-600
for v in 0 1; do
DVQ_NO_FUSED_ATTNBLOCK=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fp32-mode --no-parity --no-ae-only --no-vq-microbench 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_nofused$v.json
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench_nofused$v.json').read())
print('headline DVQ_NO_FUSED_ATTNBLOCK=$v', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms_per_step']) for k,v in d['kernel_families'].items() if 'attn' in k or 'gemm' in k})"
done
