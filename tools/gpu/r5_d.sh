#!/bin/bash
# round 5, call D: bf16-vs-fp32 loss tracking over training steps, stage-2 step with the new defaults (+ host enqueue time), the cost of
# deterministic mode on the headline step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_d}
timeout 600 python tools/debug/r5_loss_tracking.py 16 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/${TAG}_loss_tracking.txt; echo "loss tracking exit $?"; head -4 gpurun_out/${TAG}_loss_tracking.txt | cut -c1-400
timeout 400 python bench_extra.py --workload stage2 --steps 4 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_stage2.json
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_stage2.json').read())
print('stage2', d['value'], d['ms_per_step'], 'host enqueue ms/step', d.get('host_enqueue_ms_per_step'), d.get('mfma_frac_est'))"
for det in 0 1; do
DVQ_DETERMINISTIC=$det timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fp32-mode --no-parity --no-ae-only --no-vq-microbench 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_det$det.json
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench_det$det.json').read())
print('headline DVQ_DETERMINISTIC=$det', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
