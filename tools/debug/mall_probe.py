#!/usr/bin/env python3
"""Effective bandwidth of repeated device copies / read-only reductions as a function of the working set: does a working set below the
256 MB Infinity Cache stream faster than HBM?"""
import torch
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


for mb in (8, 16, 32, 64, 128, 256, 512, 1024):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    tc = timeit(lambda: y.copy_(x))
    tr = timeit(lambda: x.view(torch.int16).max())
    print(f"{mb:5d} MB tensor: copy (R+W) {2 * n * 2 / tc / 1e12:5.2f} TB/s   max-reduce (R) {n * 2 / tr / 1e12:5.2f} TB/s", flush=True)
