#!/usr/bin/env python3
"""gn_stats timings on the step's shapes whose producer does not emit the statistics itself (16x16 / 32x32 maps, attention outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (n, hw, c) in [(64, 256, 512), (64, 1024, 256), (64, 4096, 256), (64, 16384, 128), (64, 65536, 128)]:
    x = torch.randn(n, hw, c, device=dev).to(torch.bfloat16)
    us = timeit(lambda: K.gn_stats(x, 32))
    print(f"N{n} HW{hw} C{c}: {us:7.1f} us  {x.numel() * 2 / us / 1e6:5.2f} TB/s", flush=True)
