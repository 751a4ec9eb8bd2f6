#!/bin/bash
# round 5, call E: parity suite after the deterministic partial + fold path, loss tracking with controls, deterministic-mode cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_e}
rm -f gpurun_out/test_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
for det in 0 1; do
DVQ_DETERMINISTIC=$det timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fp32-mode --no-parity --no-ae-only --no-vq-microbench 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_det$det.json
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench_det$det.json').read())
print('headline DVQ_DETERMINISTIC=$det', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms_per_step']) for k,v in d['kernel_families'].items() if v['ms_per_step']>3})"
done
timeout 900 python tools/debug/r5_loss_tracking.py 16 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_loss_tracking.txt; echo "loss tracking exit $?"; tail -6 gpurun_out/${TAG}_loss_tracking.txt | cut -c1-500
