#!/bin/bash
# A/B on the GPU box: parity tests, then bench with the default kernels and with DVQ_IMPL=3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 25 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -2 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
if [ "${AB:-1}" = "1" ]; then
DVQ_IMPL=3 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vq-microbench > gpurun_out/bench_impl3.log 2> gpurun_out/bench_impl3.err; echo "bench3 exit $?"; tail -2 gpurun_out/bench_impl3.log; tail -3 gpurun_out/bench_impl3.err
fi
