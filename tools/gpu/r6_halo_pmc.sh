#!/bin/bash
# counters of the halo convolution in-step vs replayed (tools/probes/halo_instep_pmc.py), two PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
export DVQ_SIDE_WGRAD=0 MAXCAP=64
for pass in "A:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "B:TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  rm -rf "$R/gpurun_out/halopmc_$tag"
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/halopmc_$tag" -o p -- python "$R/tools/probes/halo_instep_pmc.py") > "$R/gpurun_out/halopmc_$tag.log" 2>&1
  echo "pass $tag rocprof exit $?"; tail -2 "$R/gpurun_out/halopmc_$tag.log"
  cc=$(find gpurun_out/halopmc_$tag -name "*counter_collection.csv" | head -1); kt=$(find gpurun_out/halopmc_$tag -name "*kernel_trace.csv" | head -1)
  head -1 "$kt"
  python tools/probes/halo_instep_pmc.py --summarise "$cc" "$kt" | tee gpurun_out/r06_halo_instep_pmc_$tag.txt
  rm -rf "$R/gpurun_out/halopmc_$tag"
done
