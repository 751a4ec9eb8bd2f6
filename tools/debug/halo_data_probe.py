#!/usr/bin/env python3
"""Does the dominant 3x3 conv (128->128 @256x256, B=64, bf16) run at a data- or epilogue-dependent speed?  Times the forward with
every epilogue option and the input gradient on several operand distributions (MFMA power draw, hence the clock, depends on how
many operand bits toggle)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

B, H, C = int(os.environ.get("PROBE_B", 64)), int(os.environ.get("PROBE_H", 256)), int(os.environ.get("PROBE_C", 128))
reps = int(os.environ.get("PROBE_REPS", 10))
dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
conv = Conv2d(C, C, 3, 1, 1).to(dev)
w, wt, bias = conv.packed(torch.bfloat16)
flops = 2 * B * H * H * C * C * 9


def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return round(ms, 4), round(flops / ms / 1e9)


g = torch.randn(B, H, H, C, device=dev)
data = {
    "randn": g.to(torch.bfloat16),
    "swish(randn)": (g * torch.sigmoid(g)).to(torch.bfloat16),
    "randn*1e-3": (g * 1e-3).to(torch.bfloat16),
    "zeros": torch.zeros_like(g).to(torch.bfloat16),
    "half zeros": (g * (torch.rand_like(g) > 0.5)).to(torch.bfloat16),
    "const 1": torch.ones_like(g).to(torch.bfloat16),
}
del g
r = data["randn"]
d = conv._desc(r)
K.ensure_workspace(dev)
stats = torch.zeros(B, 32, 2, dtype=torch.float64, device=dev)
for name, x in (data.items() if os.environ.get("PROBE_DATA", "1") != "0" else []):
    out = {"fwd": timeit(lambda: K.conv2d_fwd(d, x, w, bias, None)), "dgrad": timeit(lambda: K.conv2d_dgrad(d, x, wt))}
    print(f"{name:14s}", json.dumps(out), flush=True)
x = data["swish(randn)"]
print("epilogue options on swish(randn):")
print("  fwd              ", timeit(lambda: K.conv2d_fwd(d, x, w, bias, None)))
print("  fwd no bias      ", timeit(lambda: K.conv2d_fwd(d, x, w, None, None)))
print("  fwd + residual   ", timeit(lambda: K.conv2d_fwd(d, x, w, bias, r)))
print("  fwd + stats      ", timeit(lambda: K.conv2d_fwd(d, x, w, bias, None, out_stats=stats, out_groups=32)))
print("  fwd + res + stats", timeit(lambda: K.conv2d_fwd(d, x, w, bias, r, out_stats=stats, out_groups=32)))
print("  dgrad            ", timeit(lambda: K.conv2d_dgrad(d, x, wt)))
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
ss, _ = K.gn_scale_shift(r, gam, bet, 32)
print("  fwd + gn prologue          ", timeit(lambda: K.conv2d_fwd(d, r, w, bias, None, gn_ss=ss)))
print("  fwd + gn + stats           ", timeit(lambda: K.conv2d_fwd(d, r, w, bias, None, gn_ss=ss, out_stats=stats, out_groups=32)))
print("  fwd + gn + res             ", timeit(lambda: K.conv2d_fwd(d, r, w, bias, r, gn_ss=ss)))
print("  fwd + gn + res + stats     ", timeit(lambda: K.conv2d_fwd(d, r, w, bias, r, gn_ss=ss, out_stats=stats, out_groups=32)))
