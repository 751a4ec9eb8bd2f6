#!/bin/bash
# stage-2 causal attention kernels (tools/debug/attn_probe.py): timings, then PMC passes (wave / issue / LDS / memory counters)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python tools/debug/attn_probe.py 2>&1 | grep -v amdgpu.ids
REPS=2 bash tools/gpu/pmc.sh attn1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" python tools/debug/attn_probe.py | grep "attn_\|rocprof"
REPS=2 bash tools/gpu/pmc.sh attn2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" python tools/debug/attn_probe.py | grep "attn_\|rocprof"
REPS=2 bash tools/gpu/pmc.sh attn3 "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" python tools/debug/attn_probe.py | grep "attn_\|rocprof"
