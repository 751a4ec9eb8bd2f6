#!/usr/bin/env python3
"""Does the Infinity Cache serve the second read of the GroupNorm backward when the batch is processed in chunks?  Times the
backward (reduce + dx: x and dy are read twice) of a [64, 256^2, 128] bf16 tensor as one pair of launches and as per-chunk pairs
over views of 1 / 2 / 4 / 8 / 16 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
N, HW, C, G = 64, int(os.environ.get("PROBE_HW", 65536)), int(os.environ.get("PROBE_C", 128)), 32
x = torch.randn(N, HW, C, device=dev).to(torch.bfloat16)
dy = torch.randn(N, HW, C, device=dev).to(torch.bfloat16)
gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
_, mr = K.gn_forward(x, gam, bet, G, 1e-6, 1)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def chunked(k):
    for i in range(0, N, k):
        K.gn_backward(x[i:i + k], dy[i:i + k], mr[i:i + k], gam, bet, dg, db, G, 1)


for k in (64, 16, 8, 4, 2, 1):
    us = timeit(lambda: chunked(k))
    print(f"chunks of {k:2d} samples ({k * HW * C * 4 / 1e6:7.1f} MB of x + dy): {us:8.1f} us  {5 * x.numel() * 2 / us / 1e6:5.2f} TB/s by 5-pass accounting", flush=True)
