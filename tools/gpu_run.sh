#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, short bench, rocprof kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== build check"; python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
echo "== pytest -m gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf --timeout 300 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n ${PYTEST_TAIL:-60} gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -5 gpurun_out/smoke.log
if [ "${RUN_BENCH:-1}" = "1" ]; then
  echo "== bench"
  timeout ${BENCH_TIMEOUT:-600} python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
  tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1
  echo "rocprof exit $?"; cd "$OLDPWD"
  ls gpurun_out/prof/* 2>/dev/null | head; f=$(ls gpurun_out/prof/*/*kernel_stats.csv gpurun_out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
fi
