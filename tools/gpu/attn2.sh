#!/bin/bash
# round 6: second-generation causal attention (csrc/attention2.hip) -- parity tests, then timings against the first generation
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py -q -x -p no:cacheprovider --tb=short 2>&1 | tail -5
for o in 0 1; do for m in 0 1; do
  echo "== DVQ_ATTN2_ORDER=$o MASK=$m"; DVQ_ATTN2_ORDER=$o MASK=$m T=${T:-648} timeout 300 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
done; 
echo "== DVQ_ATTN2_ORDER=$o no dropout"; DVQ_ATTN2_ORDER=$o PDROP=0 T=${T:-648} timeout 300 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
echo "== DVQ_ATTN2_ORDER=$o no dropout T=2048 B=10"; DVQ_ATTN2_ORDER=$o PDROP=0 T=2048 B=10 timeout 300 python tools/debug/attn_probe.py 2>&1 | grep "fwd\|bwd"
done
