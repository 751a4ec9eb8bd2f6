#!/usr/bin/env python3
"""dvq_transpose on the shapes of a stage-2 train step: exactness against torch and bandwidth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
for (b, r, c) in [(1, 20576, 1024), (32, 643, 1024), (1, 1024, 1024), (1, 4096, 1024), (1, 1024, 4096), (3, 72, 200), (64, 1024, 256), (2, 648, 1032)]:
    x = torch.randn(b, r, c, device=dev).to(torch.bfloat16)
    y = K.transpose(x, b, r, c)
    assert torch.equal(y.view(b, c, r), x.transpose(1, 2)), (b, r, c)
    for _ in range(3): K.transpose(x, b, r, c)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): K.transpose(x, b, r, c)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"[{b}, {r}, {c}] {us:8.1f} us  {2 * x.numel() * 2 / us / 1e6:5.2f} TB/s", flush=True)
