#!/usr/bin/env python3
"""plain bf16 GEMM timings on the StackGPT shapes (M = 32 x 648 tokens): our 128x128 LDS-DMA kernel (impl 2), the 256x256
wide kernel (impl 5), auto (impl 0), and -- as a yardstick only, never a product path -- torch.matmul (hipBLASLt).
Also the TN weight-gradient GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (m, n, k) in [(20736, 1024, 1024), (20736, 3072, 1024), (20736, 4096, 1024), (20736, 1024, 4096), (8192, 8192, 8192)]:
    a2 = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b2 = torch.randn(n, k, device=dev).to(torch.bfloat16)
    a, b = a2.reshape(-1), b2.reshape(-1)
    out = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
    row = []
    for impl in (6, 7, 8, 0):
        ms = timeit(lambda: K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl))
        row.append(f"impl{impl} {ms:7.3f} ms {2.0*m*n*k/ms/1e9:6.0f} TF/s")
    ms = timeit(lambda: torch.matmul(a2, b2.t()))
    row.append(f"torch {ms:7.3f} ms {2.0*m*n*k/ms/1e9:6.0f} TF/s")
    K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=8)
    ref = torch.matmul(a2[:512].float(), b2.float().t())
    err = float((out.view(m, n)[:512].float() - ref).abs().max() / ref.abs().max())
    row.append(f"impl8 err {err:.1e}")
    print(f"NT M={m} N={n} K={k}: " + "   ".join(row), flush=True)

for (mred, i, j) in [(20736, 1024, 1024), (20736, 3072, 1024), (20736, 4096, 1024), (20736, 1024, 4096)]:
    a2 = torch.randn(mred, i, device=dev).to(torch.bfloat16)
    b2 = torch.randn(mred, j, device=dev).to(torch.bfloat16)
    out = torch.zeros(i * j, device=dev, dtype=torch.float32)
    ms = timeit(lambda: K.gemm_tn(a2.reshape(-1), b2.reshape(-1), mred, i, j, i, j, j, out=out))
    mt = timeit(lambda: torch.matmul(a2.t(), b2))
    f = 2.0 * mred * i * j
    print(f"TN Mred={mred} I={i} J={j}: ours {ms:7.3f} ms {f/ms/1e9:6.0f} TF/s   torch {mt:7.3f} ms {f/mt/1e9:6.0f} TF/s", flush=True)
