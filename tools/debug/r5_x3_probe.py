#!/usr/bin/env python3
"""What bounds the fp32x3 forward / input-gradient kernel (igemm_nt_glds_kernel<float, ., ., S3>)?  Times the 3 x 3 convolutions of the
headline step in fp32x3 with the probe library's DVQ_X3_DBG switch (wrong results): 0 = the product code, 1 = no split arithmetic (the
16-B chunks taken as ready-made bf16 planes), 2 = split, one MFMA pass of three.  Run once per setting:

    DVQ_USE_PROBES_LIB=1 DVQ_X3_DBG=<0|1|2> python tools/debug/r5_x3_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

dev = torch.device("cuda:0")
rt.set_compute_dtype("fp32x3")
K.ensure_workspace(dev)
torch.manual_seed(0)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (cin, cout, hw, n) in ((128, 128, 256, 32), (256, 256, 64, 64), (512, 512, 16, 64)):
    conv = Conv2d(cin, cout, 3, stride=1, padding=1).to(dev)
    g = torch.randn(n, hw, hw, cin, device=dev)
    x = g * torch.sigmoid(g)
    d = conv._desc(x)
    w, wt, bias = conv.packed(torch.float32)
    dy = torch.randn(n, hw, hw, cout, device=dev)
    fl = 2.0 * n * hw * hw * cout * 9 * cin
    tf, td = timeit(lambda: K.conv2d_fwd(d, x, w, bias)), timeit(lambda: K.conv2d_dgrad(d, dy, wt))
    print(f"DVQ_X3_DBG={os.environ.get('DVQ_X3_DBG', '0')}  3x3 {cin}->{cout} @{hw} N{n}: fwd {tf:.3f} ms {fl / tf / 1e9:.0f} TF/s   dgrad {td:.3f} ms {fl / td / 1e9:.0f} TF/s")
