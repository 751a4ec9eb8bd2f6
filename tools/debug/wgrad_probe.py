#!/usr/bin/env python3
"""3x3 halo weight-gradient kernel timings on the step's shapes (bf16, B = 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d
dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
K.ensure_workspace(dev)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


SHAPES = [(64, 256, 128, 128, 10), (64, 128, 128, 128, 9), (64, 64, 256, 256, 9), (64, 32, 256, 256, 20), (64, 128, 256, 128, 1), (64, 256, 128, 8, 3)]
for (n, h, cin, cout, per_step) in SHAPES[:int(os.environ.get('PROBE_NSHAPES', 6))]:
    conv = Conv2d(cin, cout, 3, 1, 1).to(dev)
    x = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
    cout_p = -(-cout // 8) * 8
    dy = torch.randn(n, h, h, cout_p, device=dev).to(torch.bfloat16)
    d = conv._desc(x)
    gw = torch.zeros(cout, 3, 3, cin, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
    gb = torch.zeros(cout, dtype=torch.float32, device=dev)
    ms = timeit(lambda: K.conv2d_wgrad_oihw(d, x, dy, cin, cout, gw, gb))
    fl = 2.0 * n * h * h * cin * cout_p * 9
    print(f"wgrad N{n} {h}x{h} {cin}->{cout}: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF/s  x{per_step} = {ms * per_step:6.2f} ms/step", flush=True)
