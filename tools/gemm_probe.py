#!/usr/bin/env python3
"""plain bf16 NT GEMM timings: 128x128 kernel (impl 2) vs auto (256x256 wide kernel when eligible)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
for (m, n, k) in [(20736, 1024, 1024), (20736, 4096, 1024), (20736, 1024, 4096), (20736, 1032, 1024), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16).reshape(-1)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16).reshape(-1)
    out = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
    row = []
    for impl in (2, 0):
        for _ in range(2):
            K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        row.append(f"impl{impl} {ms:7.3f} ms {2.0*m*n*k/ms/1e9:6.0f} TF/s")
    print(f"M={m} N={n} K={k}: " + "   ".join(row), flush=True)
