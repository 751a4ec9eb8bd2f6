#!/bin/bash
# round 5: fp32x3 weight gradients on bf16 planes -- tests, same-box A/B of the fp32x3 step, per-shape table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split" -p no:cacheprovider 2>&1 | tail -5
B="--dtype fp32x3 --steps 6 --warmup 2 --no-extras --no-fp32-mode --no-parity --no-cpu-baseline --no-ae-only --no-vq-microbench"
for m in 1 0 1 0; do
  DVQ_X3_WGRAD_PLANES=$m timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('planes=$m', d['value'], d['ms_per_step'])"
done
DVQ_SHAPES_DTYPE=fp32x3 DVQ_SIDE_WGRAD=0 TOP=70 timeout 300 python tools/debug/step_shapes.py 2>/dev/null | grep -v "Warn\|return get_obj" > gpurun_out/r5_x3_step_shapes_planes.txt; head -12 gpurun_out/r5_x3_step_shapes_planes.txt
