#!/usr/bin/env python3
"""How well do an HBM-bound GroupNorm pass and an MFMA-bound convolution kernel share the chip when launched on two streams?
Times each alone and both together (128 channels, 256^2, B = 64, bf16): together ~ max(alone) means the GroupNorm pass hides behind
the convolution, together ~ sum means the two-stream schedule buys nothing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d
dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
K.ensure_workspace(dev)
B, H, C = int(os.environ.get("PROBE_B", 64)), int(os.environ.get("PROBE_H", 256)), 128
reps = 10
conv = Conv2d(C, C, 3, 1, 1).to(dev)
w, wt, bias = conv.packed(torch.bfloat16)
x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
dy = torch.randn_like(x)
x2 = torch.randn_like(x)
dy2 = torch.randn_like(x)
d = conv._desc(x)
gw = torch.zeros(C, 3, 3, C, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
gb = torch.zeros(C, dtype=torch.float32, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
_, mr = K.gn_forward(x2, gamma, beta)
stats = K.gn_stats(x2, 32)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

mfma = {
    "conv fwd": lambda: K.conv2d_fwd(d, x, w, bias, None),
    "conv wgrad": lambda: K.conv2d_wgrad_oihw(d, x, dy, C, C, gw, gb),
}
hbm = {
    "gn_apply": lambda: K.gn_forward(x2, gamma, beta, stats=stats),
    "gn_backward": lambda: K.gn_backward(x2, dy2, mr, gamma, beta, dg, db),
    "copy": lambda: dy2.copy_(x2),
}


def run(fm, fh):
    """reps launches of fm on the main stream and of fh on the side stream (either may be None); wall time per pair"""
    for _ in range(2):
        if fm: fm()
        if fh:
            with torch.cuda.stream(side): fh()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(main)
    side.wait_event(s)
    for _ in range(reps):
        if fm: fm()
        if fh:
            with torch.cuda.stream(side): fh()
    j = torch.cuda.Event()
    j.record(side)
    main.wait_event(j)
    e.record(main)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for mn, fm in mfma.items():
    tm = run(fm, None)
    for hn, fh in hbm.items():
        th = run(None, fh)
        tb = run(fm, fh)
        print(f"{mn:11s} {tm:6.3f} ms | {hn:12s} {th:6.3f} ms | together {tb:6.3f} ms  (sum {tm + th:6.3f}, max {max(tm, th):6.3f}, hidden {100 * (tm + th - tb) / min(tm, th):5.1f} % of the shorter)", flush=True)
