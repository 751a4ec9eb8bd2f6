#!/usr/bin/env python3
"""NT GEMM: the 8-phase kernel (impl 10) against the pipelined wide kernel (impl 6 / 8), auto (impl 0) and -- yardstick only, never a
product path -- torch.matmul (hipBLASLt), on uniform random data; full-tensor check of impl 10 against an fp32 torch product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
impls = [int(v) for v in os.environ.get("IMPLS", "6,10,0").split(",")]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (20736, 1024, 1024), (20736, 3072, 1024), (20736, 4096, 1024), (20736, 1024, 4096),
          (20576, 1024, 1024), (1000, 520, 192), (256, 256, 64), (300, 264, 128)]
for (m, n, k) in shapes:
    torch.manual_seed(m + n + k)
    a2 = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
    b2 = (torch.rand(n, k, device=dev) * 2 - 1).to(torch.bfloat16)
    a, b = a2.reshape(-1), b2.reshape(-1)
    out = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
    row = []
    for impl in impls:
        try:
            ms = timeit(lambda: K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=impl))
            row.append(f"impl{impl} {ms:7.3f} ms {2.0*m*n*k/ms/1e9:6.0f} TF/s")
        except Exception as e:
            row.append(f"impl{impl} failed: {str(e)[:60]}")
    ms = timeit(lambda: torch.matmul(a2, b2.t()))
    row.append(f"torch {ms:7.3f} ms {2.0*m*n*k/ms/1e9:6.0f} TF/s")
    out.zero_()
    K.gemm_nt(a, b, m, n, k, k, k, n, out=out, impl=int(os.environ.get("CHECK_IMPL", "10")))
    worst = 0.0
    for r0 in range(0, m, 4096):
        ref = torch.matmul(a2[r0:r0 + 4096].float(), b2.float().t())
        got = out.view(m, n)[r0:r0 + 4096].float()
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    row.append(f"checked-impl full-tensor err {worst:.1e}")
    print(f"NT M={m} N={n} K={k}: " + "   ".join(row), flush=True)
# repeatability (races show up as run-to-run differences)
m, n, k = 4096, 4096, 4096
a2 = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
b2 = (torch.rand(n, k, device=dev) * 2 - 1).to(torch.bfloat16)
outs = []
for _ in range(20):
    o = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
    K.gemm_nt(a2.reshape(-1), b2.reshape(-1), m, n, k, k, k, n, out=o, impl=int(os.environ.get("CHECK_IMPL", "10")))
    outs.append(o)
torch.cuda.synchronize()
print("repeatability: differing runs", sum(int(not torch.equal(outs[0], o)) for o in outs[1:]), "of 19")
