#!/usr/bin/env python3
"""Output statistics of the halo convolution (sum, sum of squares per (image, group) of the bf16 values AS STORED) against a float64
evaluation of the stored tensor: the matrix-pipe path (default) and the vector path (DVQ_HALO_MFMA_STATS=0), run in two processes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d
dev = torch.device("cuda:0")
torch.manual_seed(3)
for (n, h, w, cin, cout, res) in [(4, 64, 64, 128, 128, False), (4, 64, 64, 128, 128, True), (2, 32, 64, 256, 256, False)]:
    with rt.compute_dtype_ctx(torch.bfloat16):
        conv = Conv2d(cin, cout, 3, stride=1, padding=1).to(dev)
        x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
        r = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16) if res else None
        d = conv._desc(x)
        wp, wt, bias = conv.packed(torch.bfloat16)
        K.ensure_workspace(dev)
        st = torch.zeros(n, 32, 2, dtype=torch.float64, device=dev)
        y = K.conv2d_fwd(d, x, wp, bias, r, out_stats=st, out_groups=32)
        torch.cuda.synchronize()
        yd = y.double().view(n, h * w, 32, cout // 32)
        ref = torch.stack([yd.sum((1, 3)), (yd * yd).sum((1, 3))], -1)
        err = ((st - ref).abs() / ref.abs().clamp_min(1e-30))
        # relative to the sum of squares scale for the plain sums (they cancel)
        e1 = float(((st[..., 0] - ref[..., 0]).abs() / ref[..., 1].sqrt().clamp_min(1e-30) / (h * w * cout / 32) ** 0.5).max())
        e2 = float(err[..., 1].max())
        print(f"N{n} {h}x{w} {cin}->{cout} res={res}: sum err / (rms * sqrt(count)) {e1:.2e}   sum-of-squares rel err {e2:.2e}")
