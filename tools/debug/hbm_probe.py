#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box for the GroupNorm-shaped streams: device copy (1 read + 1 write), fill (write only) and the
library's gn_stats (read only) / gn_apply (read + write) / gn backward (4 reads + 1 write) on a 1.07 GB bf16 tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


B, H, C = 64, 256, 128
x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
dy = torch.randn_like(x)
y = torch.empty_like(x)
gb = x.numel() * 2 / 1e9
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
_, mr = K.gn_forward(x, gamma, beta)
stats = K.gn_stats(x, 32)
rows = [
    ("copy_ (1R + 1W)", lambda: y.copy_(x), 2),
    ("fill_ (1W)", lambda: y.fill_(1.0), 1),
    ("gn_stats (1R)", lambda: K.gn_stats(x, 32), 1),
    ("gn_apply (1R + 1W)", lambda: K.gn_forward(x, gamma, beta, stats=stats), 2),
    ("gn_backward (4R + 1W)", lambda: K.gn_backward(x, dy, mr, gamma, beta, dg, db), 5),
]
for name, fn, passes in rows:
    ms = timeit(fn)
    print(f"{name:24s} {ms:7.3f} ms   {passes * gb / ms:6.2f} TB/s")
