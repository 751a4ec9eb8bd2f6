#!/usr/bin/env python3
"""Main-loop timing splits of the pipelined 256 x 256 GEMM (DVQ_GEMM_DBG bits: 1 re-read slab 0, 2 print cycles per slab, 4 no
slab barrier, 8 no DMA after slab 0 -- results are wrong with 4 / 8, timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
m, n, k = [int(v) for v in os.environ.get("MNK", "8192,8192,8192").split(",")]
a = torch.randn(m, k, device=dev).to(torch.bfloat16).reshape(-1)
b = torch.randn(n, k, device=dev).to(torch.bfloat16).reshape(-1)
out = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
for impl in (6, 7):
    for _ in range(2): K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"impl{impl} {ms:.3f} ms {2.0*m*n*k/ms/1e9:.0f} TF/s", flush=True)
