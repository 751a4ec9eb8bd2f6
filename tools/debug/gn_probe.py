#!/usr/bin/env python3
"""GroupNorm / BatchNorm pass timings (HBM-bound): apply, backward reduce (with per-block partials or atomics), backward dx."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (n, h, c, g, act) in [(64, 256, 128, 32, 1), (64, 128, 128, 32, 1), (64, 128, 256, 32, 1), (64, 64, 256, 32, 1), (64, 32, 256, 32, 1), (64, 16, 512, 32, 1),
                          (1, 64 * 64 * 64, 128, 128, 2), (1, 64 * 32 * 32, 256, 256, 2)]:
    hw = h * h if n > 1 else h
    x = torch.randn(n, hw, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, hw, c, device=dev).to(torch.bfloat16)
    gam, bet = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    y, mr = K.gn_forward(x, gam, bet, g, 1e-6, act)
    nb = x.numel() * 2
    row = [f"N{n} HW{hw} C{c} G{g}"]
    row.append(f"apply {timeit(lambda: K.gn_forward(x, gam, bet, g, 1e-6, act)):7.1f} us")
    for flag in (True, False):
        K._GN_PARTIALS = flag
        us = timeit(lambda: K.gn_backward(x, dy, mr, gam, bet, dg, db, g, act))
        row.append(f"bwd({'partials' if flag else 'atomics'}) {us:7.1f} us {5 * nb / us / 1e6:5.2f} TB/s")
    # the two passes of the backward on their own (partials path)
    import ctypes as C
    from dynamicvectorquantization_amd.kernels import lib, _p, _s, dt, check
    K._GN_PARTIALS = True
    red = torch.zeros(n, g, 2, dtype=torch.float64, device=dev)
    part = torch.empty(lib().dvq_gn_bwd_partial_bytes(n, hw, c), dtype=torch.uint8, device=dev)
    dxb = torch.empty_like(x)
    us_r = timeit(lambda: check(lib().dvq_gn_bwd_reduce(_p(x), _p(dy), dt(x), n, hw, c, g, _p(mr), _p(gam), _p(bet), int(act), _p(red), _p(dg), _p(db), _p(part), _s()), "r"))
    us_d = timeit(lambda: check(lib().dvq_gn_bwd_dx(_p(x), _p(dy), dt(x), n, hw, c, g, _p(mr), _p(gam), _p(bet), int(act), _p(red), None, _p(dxb), _s()), "d"))
    row.append(f"reduce alone {us_r:7.1f} us  dx alone {us_d:7.1f} us")
    K._GN_PARTIALS = True
    dg.zero_(); db.zero_()
    d1 = K.gn_backward(x, dy, mr, gam, bet, dg, db, g, act); g1, b1 = dg.clone(), db.clone()
    K._GN_PARTIALS = False
    dg.zero_(); db.zero_()
    d0 = K.gn_backward(x, dy, mr, gam, bet, dg, db, g, act)
    row.append(f"dx diff {float((d1.float() - d0.float()).abs().max()):.1e} dgamma rel {float((g1 - dg).abs().max() / dg.abs().max()):.1e} dbeta rel {float((b1 - db).abs().max() / db.abs().max()):.1e}")
    print("  ".join(row), flush=True)
