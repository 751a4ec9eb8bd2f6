#!/bin/bash
# generic GPU-box launcher: runs the given command line(s) from the repo root with the usual environment, logs under gpurun_out/
# usage: gpurun -- 'bash tools/gpu/run.sh <tag> "<cmd1>" "<cmd2>" ...'   -> gpurun_out/<tag>_<i>.log
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift; i=0
for c in "$@"; do
  i=$((i+1))
  echo "=== [$tag $i] $c"
  bash -c "$c" > "gpurun_out/${tag}_$i.log" 2>&1; echo "exit $?"
  tail -n 3 "gpurun_out/${tag}_$i.log" | cut -c1-300
done
