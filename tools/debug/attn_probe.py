#!/usr/bin/env python3
"""stage-2 causal attention kernels at the p6c18 geometry (B = 32, T = 648, 8 heads of 128, dropout 0.1): forward / backward timings"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
b, t, nh, hs = int(os.environ.get("B", 32)), int(os.environ.get("T", 648)), 8, 128
c = nh * hs
p_drop = float(os.environ.get("PDROP", 0.1))
reps = int(os.environ.get("REPS", 10))
q, k, v, do = [torch.randn(b * t, c, device=dev).to(torch.bfloat16) for _ in range(4)]
scale = hs ** -0.5


def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


dm = K.attn_causal_drop_mask(q, b, t, nh) if p_drop > 0 and os.environ.get("MASK", "1") == "1" else None
out, lse = K.attn_causal_fwd(q, k, v, b, t, nh, scale, p_drop, 7, drop_mask=dm)
fl = 2.0 * b * t * t * c            # 4 B nh T^2/2 hs
ms = timeit(lambda: K.attn_causal_fwd(q, k, v, b, t, nh, scale, p_drop, 7, drop_mask=dm))
print(f"fwd  {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF/s")
ms = timeit(lambda: K.attn_causal_bwd(q, k, v, out, do, lse, b, t, nh, scale, p_drop, 7, drop_mask=dm))
print(f"bwd  {ms:7.3f} ms {2.5 * fl / ms / 1e9:6.0f} TF/s   (x24 layers = {ms * 24:6.2f} ms/step)")
