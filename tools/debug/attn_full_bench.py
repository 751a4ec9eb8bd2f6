#!/usr/bin/env python3
"""AttnBlock attention (one head of 256 channels over T tokens, B = 64): flash forward and the backward path, timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import _lib, kernels as K
dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
b, t, c = 64, int(os.environ.get("T", "1024")), 256
reps = int(os.environ.get("REPS", "10"))
q, k, v, do = (torch.randn(b * t, c, device=dev).to(torch.bfloat16) for _ in range(4))


def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


o, lse = K.attn_full_fwd(q, k, v, b, t, c ** -0.5)
fl = 4.0 * b * t * t * c
ms = timeit(lambda: K.attn_full_fwd(q, k, v, b, t, c ** -0.5))
print(f"attn_full_fwd  {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF/s")
ms = timeit(lambda: K.attn_full_bwd(q, k, v, o, do, lse, b, t, c ** -0.5))
print(f"attn_full_bwd (flash kernels) {ms:7.3f} ms {2.5 * fl / ms / 1e9:6.0f} TF/s")
