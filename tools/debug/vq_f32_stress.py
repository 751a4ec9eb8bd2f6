"""is the fp32-rows VQ kernel deterministic and exact at the small shapes the stage-2 tests use?  (N = 2048 / 4096 / 65536, D = 64 / 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamicvectorquantization_amd import kernels as K
from oracle import vq as ovq
dev = torch.device("cuda:0")
for (n, d, k) in ((2048, 64, 512), (4096, 64, 512), (4224, 64, 512), (65536, 256, 1024)):
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g).to(dev)
    cb = (torch.randn(k, d, generator=g) * 0.7).to(dev)
    prep = K.vq_prepare(cb)
    ref = None
    bad = 0
    for rep in range(40):
        idx = K.vq_argmin(x, cb, prep)
        if ref is None:
            ref = idx.clone()
            if n <= 4224:
                want = ovq.argmin_exact(x.cpu().numpy(), cb.cpu().numpy())
                print(n, d, k, "vs exact oracle mismatches:", int((ref.cpu().numpy() != want).sum()))
        else:
            bad += int((idx != ref).sum())
    torch.cuda.synchronize()
    print(n, d, k, "run-to-run differing indices over 39 repeats:", bad)
