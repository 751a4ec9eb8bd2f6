#!/bin/bash
# round-3 baseline: per-shape timing of one headline step + one-rank DP replay cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TOP=140 timeout 400 python tools/debug/step_shapes.py > gpurun_out/r3_step_shapes_base.txt 2> gpurun_out/r3_step_shapes_base.err; echo "shapes exit $?"
timeout 500 python bench.py --steps 10 --warmup 2 --mode graph --no-cpu-baseline --no-ae-only > gpurun_out/r3_base_graph.json 2> gpurun_out/r3_base_graph.err; echo "graph exit $?"
DVQ_FORCE_DP=1 timeout 500 python bench.py --steps 10 --warmup 2 --mode graph --no-cpu-baseline --no-ae-only --no-vq-microbench > gpurun_out/r3_base_dp.json 2> gpurun_out/r3_base_dp.err; echo "dp exit $?"
tail -c 600 gpurun_out/r3_base_graph.json; echo; tail -c 600 gpurun_out/r3_base_dp.json
