#!/bin/bash
# round 6: second-generation attention (csrc/attention2.hip) -- parity tests, then timings against the first generation
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for pipe in 0 1; do
echo "=== DVQ_ATTN2_PIPE=$pipe"
DVQ_ATTN2_PIPE=$pipe timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider --tb=short -k "attn or Attn or attention" 2>&1 | tail -5
for m in 1; do
  echo "== causal 128, dropout, MASK=$m"; DVQ_ATTN2_PIPE=$pipe MASK=$m T=${T:-648} timeout 300 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
done
echo "== causal 128, no dropout"; DVQ_ATTN2_PIPE=$pipe PDROP=0 T=${T:-648} timeout 300 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
echo "== full 256 (AttnBlock), v2"; DVQ_ATTN2_PIPE=$pipe timeout 300 python tools/debug/attn_full_bench.py 2>&1 | grep "attn_full\|AttnBlock\|not timed"
done
echo "== full 256 (AttnBlock), first generation"; DVQ_ATTN_V2=0 timeout 300 python tools/debug/attn_full_bench.py 2>&1 | grep "attn_full\|AttnBlock\|not timed"
