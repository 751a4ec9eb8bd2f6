#!/usr/bin/env python3
"""Micro-benchmark of the fused single-head AttnBlock attention (B=64, T=1024, C=256) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import _lib, kernels as K
dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
b, t, c = 64, int(os.environ.get("T", "1024")), 256
q, k, v, do = (torch.randn(b * t, c, device=dev).to(torch.bfloat16) for _ in range(4))
for _ in range(int(os.environ.get("REPS", "10"))):
    o, lse = K.attn_full_fwd(q, k, v, b, t, c ** -0.5)
    K.attn_full_bwd(q, k, v, o, do, lse, b, t, c ** -0.5)
torch.cuda.synchronize()
