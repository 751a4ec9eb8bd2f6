#!/usr/bin/env python3
"""What do the fused GroupNorm pieces of the 3x3 halo kernel cost?  128->128 @ 256x256, N=64 (and 128x128): plain / +GN-swish
prologue (gn_ss) / +output statistics (out_stats) / both / both + residual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
for (n, h, ci, co) in [(64, 256, 128, 128), (64, 128, 128, 128), (64, 64, 256, 256)]:
    conv = Conv2d(ci, co, 3, 1, 1).to(dev)
    x = torch.randn(n, h, h, ci, device=dev).to(torch.bfloat16)
    res = torch.randn(n, h, h, co, device=dev).to(torch.bfloat16)
    d = conv._desc(x)
    w, wt, bias = conv.packed(torch.bfloat16)
    ss = torch.rand(n, ci, 2, device=dev, dtype=torch.float32)
    flops = 2.0 * n * h * h * ci * co * 9
    row = []
    for tag, kw in (("plain", {}), ("gn_ss", dict(gn_ss=ss)), ("stats", dict(stats=True)), ("both", dict(gn_ss=ss, stats=True)),
                    ("both+res", dict(gn_ss=ss, stats=True, residual=res))):
        def fn():
            st = torch.zeros(n, 32, 2, device=dev, dtype=torch.float64) if kw.get("stats") else None
            return K.conv2d_fwd(d, x, w, bias, residual=kw.get("residual"), gn_ss=kw.get("gn_ss"), out_stats=st, out_groups=32 if st is not None else 0)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        row.append(f"{tag} {ms:6.3f} ms {flops / ms / 1e9:5.0f}")
    print(f"N={n} {h}x{h} {ci}->{co}: " + " | ".join(row), flush=True)
