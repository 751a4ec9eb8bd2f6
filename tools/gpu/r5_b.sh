#!/bin/bash
# round 5, call B: full parity suite on the new kernels (per-code VQ bound, Linear multi-pack, LayerNorm-backward dropout output,
# concurrent sampling lanes), the fixed precision probe (+ the no-warm-up VQ operands), per-shape table, stage-2 / sampling A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_b}
rm -f gpurun_out/test_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 6 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
cp gpurun_out/test_reports.jsonl gpurun_out/${TAG}_test_reports.jsonl 2>/dev/null
timeout 200 python bench.py --vq-only > gpurun_out/${TAG}_vq_only.json 2>/dev/null; echo "vq-only exit $?"; python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_vq_only.json').read().strip().splitlines()[-1])['vq_argmin']
print({k:(v['ms'],v['rerank_rows_candidates'],v['rerank_rows_wide']) for k,v in d.items()})"
for m in "1 1" "0 1" "1 0"; do set -- $m
  DVQ_LINEAR_MULTIPACK=$1 DVQ_FUSE_DROP_BWD=$2 timeout 400 python bench_extra.py --workload stage2 --steps 4 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_stage2_mp$1_fd$2.json
  python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_stage2_mp$1_fd$2.json').read())
print('stage2 multipack=$1 fuse_drop=$2', d['value'], d['ms_per_step'], d.get('mfma_frac_est'))"
done
timeout 500 python bench_extra.py --workload sampling --no-cpu-baseline 2>gpurun_out/${TAG}_sampling.err | tail -1 > gpurun_out/${TAG}_sampling.json; echo "sampling exit $?"
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_sampling.json').read())
print({k:v['token_steps_per_sec'] for k,v in d['by_batch'].items()}, {k:(v.get('token_steps_per_sec'), v.get('failed')) for k,v in d.get('by_batch_concurrent_lanes',{}).items()})"
timeout 600 python tools/debug/r5_precision_probe.py 30 > gpurun_out/${TAG}_precision_probe.txt 2>&1; echo "probe exit $?"; grep "^mixed\|^all-bf16" gpurun_out/${TAG}_precision_probe.txt | cut -c1-300
DVQ_PROBE_WARMUP0=1 timeout 300 python tools/debug/r5_precision_probe.py 30 vq_only > gpurun_out/${TAG}_vq_probe_warmup0.txt 2>&1; echo "probe warmup0 exit $?"; grep "vq_in_training" gpurun_out/${TAG}_vq_probe_warmup0.txt | cut -c1-1800
DVQ_SIDE_WGRAD=0 TOP=160 timeout 300 python tools/debug/step_shapes.py 2>gpurun_out/${TAG}_step_shapes.err | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_step_shapes.txt; echo "shapes exit $?"; tail -3 gpurun_out/${TAG}_step_shapes.err | cut -c1-300
