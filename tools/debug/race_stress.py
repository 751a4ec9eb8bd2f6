#!/usr/bin/env python3
"""Determinism stress of the software-pipelined kernels: every launch of a kernel on the same operands must reproduce the first
result bit for bit (an LDS race -- DMA landing after a buffer was re-used, a missing barrier -- shows up as a handful of differing
elements in one launch out of a few).  Full-size shapes, REPS launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
REPS = int(os.environ.get("REPS", "40"))
bad = 0


def check(name, fn):
    global bad
    ref = [t.clone() for t in fn()]
    nd = 0
    for _ in range(REPS):
        out = fn()
        nd += sum(int((a != b).sum()) for a, b in zip(out, ref))
    print(f"{name:58s} {'OK' if nd == 0 else f'{nd} differing elements'}", flush=True)
    bad += nd != 0


K.ensure_workspace(dev)
for (B, H, Cin, Cout) in [(64, 256, 128, 128), (64, 64, 256, 256), (64, 32, 512, 512), (16, 128, 64, 256)]:
    conv = Conv2d(Cin, Cout, 3, 1, 1).to(dev)
    w, wt, bias = conv.packed(torch.bfloat16)
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    r = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    d = conv._desc(x)
    g = 32 if Cout % 32 == 0 else 0

    def fwd():
        st = torch.zeros(B, 32, 2, dtype=torch.float64, device=dev)
        y = K.conv2d_fwd(d, x, w, bias, r, out_stats=st if g else None, out_groups=g)
        return [y, st]
    check(f"halo fwd+res+stats  B{B} {H}x{H} {Cin}->{Cout}", fwd)
    check(f"halo dgrad          B{B} {H}x{H} {Cin}->{Cout}", lambda: [K.conv2d_dgrad(d, dy, wt)])

    def wg():
        gw = torch.zeros(Cout, 3, 3, Cin, dtype=torch.float32, device=dev)
        gb = torch.zeros(Cout, dtype=torch.float32, device=dev)
        K.conv2d_wgrad_oihw(d, x, dy, Cin, Cout, gw, gb)
        return [gw, gb]
    check(f"halo wgrad          B{B} {H}x{H} {Cin}->{Cout}", wg)
    del x, r, dy

for (m, n, k) in [(20736, 1024, 1024), (20736, 4096, 1024), (20736, 1024, 4096), (65536, 256, 256), (4100, 520, 192)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16).reshape(-1)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16).reshape(-1)
    bias = torch.randn(n, device=dev)
    for impl in (6, 8):
        check(f"pipelined NT GEMM impl {impl}  {m}x{n}x{k}", lambda: [K.gemm_nt(a, b, m, n, k, k, k, n, bias=bias, bias_mode=1, impl=impl).clone()])
for (mred, i, j) in [(20736, 1024, 1024), (20736, 4096, 1024), (65536, 256, 256), (5000, 520, 264)]:
    a = torch.randn(mred, i, device=dev).to(torch.bfloat16).reshape(-1)
    b = torch.randn(mred, j, device=dev).to(torch.bfloat16).reshape(-1)
    check(f"pipelined TN GEMM (workspace fold)  {mred}x{i}x{j}", lambda: [K.gemm_tn(a, b, mred, i, j, i, j, j)])
print("FAILED" if bad else "all deterministic")
sys.exit(1 if bad else 0)
