#!/usr/bin/env python3
"""Isolated timings of the convolution shapes of one headline train step that do NOT run on the 3x3 halo kernel (strided, 4x4,
1x1 and small-map convolutions: SURVEY 8(a) rows a6-a10, a15), forward / input gradient / weight gradient, each checked against
the naive direct kernel (impl 1) on a reduced batch.  One line per (shape, pass): microseconds, TFLOP/s, algorithmic GB/s.

    python tools/conv_bench.py [--only fwd,dgrad,wgrad] [--impl 0] [--filter 4x4] [--reps 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

# (label, cin, cout, k, stride, pad, asym, H, W, N, launches per step fwd / dgrad / wgrad)
SHAPES = [
    ("1x1 256->256 @32", 256, 256, 1, 1, 0, False, 32, 32, 64, (60, 30, 30)),
    ("1x1 512->512 @16", 512, 512, 1, 1, 0, False, 16, 16, 64, (24, 12, 12)),
    ("1x1 256->128 @128", 256, 128, 1, 1, 0, False, 128, 128, 64, (2, 1, 1)),
    ("1x1 128->256 @64", 128, 256, 1, 1, 0, False, 64, 64, 64, (2, 1, 1)),
    ("3x3 512->512 @16", 512, 512, 3, 1, 1, False, 16, 16, 64, (14, 10, 7)),
    ("3x3 256->512 @16", 256, 512, 3, 1, 1, False, 16, 16, 64, (2, 1, 1)),
    ("3x3 512->256 @16", 512, 256, 3, 1, 1, False, 16, 16, 64, (2, 1, 1)),
    ("3x3s2 128->128 @256", 128, 128, 3, 2, 0, True, 256, 256, 64, (2, 1, 1)),
    ("3x3s2 128->128 @128", 128, 128, 3, 2, 0, True, 128, 128, 64, (2, 1, 1)),
    ("3x3s2 256->256 @64", 256, 256, 3, 2, 0, True, 64, 64, 64, (2, 1, 1)),
    ("3x3s2 256->256 @32", 256, 256, 3, 2, 0, True, 32, 32, 64, (2, 1, 1)),
    ("4x4s2 64->128 @128", 64, 128, 4, 2, 1, False, 128, 128, 64, (3, 3, 2)),
    ("4x4s2 128->256 @64", 128, 256, 4, 2, 1, False, 64, 64, 64, (3, 3, 2)),
    ("4x4s1 256->512 @32", 256, 512, 4, 1, 1, False, 32, 32, 64, (3, 3, 2)),
    ("4x4s1 512->8 @31", 512, 8, 4, 1, 1, False, 31, 31, 64, (3, 3, 2)),
    ("4x4s2 8->64 @256", 8, 64, 4, 2, 1, False, 256, 256, 64, (3, 1, 2)),
    ("3x3 8->128 @256", 8, 128, 3, 1, 1, False, 256, 256, 64, (2, 1, 1)),
    ("3x3 128->8 @256", 128, 8, 3, 1, 1, False, 256, 256, 64, (2, 1, 3)),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3           # microseconds


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--filter", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    passes = a.only.split(",")
    dev = torch.device("cuda:0")
    rt.set_compute_dtype(torch.bfloat16)
    K.ensure_workspace(dev)
    torch.manual_seed(0)
    tot = {p: 0.0 for p in passes}
    for (label, cin, cout, k, s, pad, asym, h, w_, n, per_step) in SHAPES:
        if a.filter and a.filter not in label:
            continue
        conv = Conv2d(cin, cout, k, stride=s, padding=pad, asym_pad=asym).to(dev)
        g = torch.randn(n, h, w_, cin, device=dev)
        x = (g * torch.sigmoid(g)).to(torch.bfloat16)
        del g
        with rt.impl_ctx(a.impl):
            d = conv._desc(x)
        w, wt, bias = conv.packed(torch.bfloat16)
        dy = torch.randn(n, d.OH, d.OW, d.Cout, device=dev).to(torch.bfloat16)
        gw = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)      # OHWI storage like FlatParams
        gb = torch.zeros(cout, dtype=torch.float32, device=dev)
        flops = 2.0 * n * d.OH * d.OW * d.Cout * k * k * d.Cin
        nbytes = 2.0 * (x.numel() + dy.numel())
        fns = {"fwd": lambda: K.conv2d_fwd(d, x, w, bias), "dgrad": lambda: K.conv2d_dgrad(d, dy, wt),
               "wgrad": lambda: K.conv2d_wgrad_oihw(d, x, dy, cin, cout, gw, gb)}
        # reference on a reduced batch: naive direct kernels
        nr = min(n, 4)
        with rt.impl_ctx(1):
            d1 = conv._desc(x[:nr])
        with rt.impl_ctx(a.impl):
            dr = conv._desc(x[:nr])
        for pi, ps in enumerate(("fwd", "dgrad", "wgrad")):
            if ps not in passes:
                continue
            err = float("nan")
            if not a.no_check:
                xs, dys = x[:nr].contiguous(), dy[:nr].contiguous()
                if ps == "fwd":
                    err = relerr(K.conv2d_fwd(dr, xs, w, bias), K.conv2d_fwd(d1, xs, w, bias))
                elif ps == "dgrad":
                    err = relerr(K.conv2d_dgrad(dr, dys, wt), K.conv2d_dgrad(d1, dys, wt))
                else:
                    ga = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
                    gr_ = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
                    ba, br = torch.zeros_like(gb), torch.zeros_like(gb)
                    K.conv2d_wgrad_oihw(dr, xs, dys, cin, cout, ga, ba)
                    K.conv2d_wgrad_oihw(d1, xs, dys, cin, cout, gr_, br)
                    err = max(relerr(ga, gr_), relerr(ba, br))
            us = timeit(fns[ps], a.reps)
            tot[ps] += us * per_step[pi] / 1e3
            print(f"{label:22s} {ps:5s} {us:9.1f} us {flops / us / 1e6:7.0f} TF/s {nbytes / us / 1e3:7.0f} GB/s  x{per_step[pi]:2d} "
                  f"= {us * per_step[pi] / 1e3:6.3f} ms/step  err {err:.1e}", flush=True)
        del x, dy
    print("per-step totals (ms):", {k_: round(v, 3) for k_, v in tot.items()}, "sum", round(sum(tot.values()), 3))


if __name__ == "__main__":
    main()
