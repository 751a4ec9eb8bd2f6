#!/usr/bin/env python3
"""Determinism stress of the fp32x3 halo path (dvq_conv2d_fwd_x3 / dvq_conv2d_dgrad_x3: pre-pass, weight re-layout, halo main loop on
3 Cin channels, fp32-output epilogue): forward (+ residual, + ReLU) and input gradient (+ gate) have no atomics, so every launch on the
same operands must reproduce the first result bit for bit.  NPROC processes share the GPU (round 4's LDS-DMA race only showed once a
second process competed for the CUs); full-size shapes, REPS launches each.

    python tools/debug/r5_x3_race_stress.py [nproc=2]        REPS=30"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank):
    import torch
    from dynamicvectorquantization_amd import kernels as K, runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    dev = torch.device("cuda:0")
    rt.set_compute_dtype("fp32x3")
    torch.manual_seed(rank)
    reps = int(os.environ.get("REPS", "30"))
    bad = 0

    def check(name, fn):
        nonlocal bad
        ref = fn().clone()
        nd = 0
        for _ in range(reps):
            nd += int((fn() != ref).sum())
        print(f"[{rank}] {name:52s} {'OK' if nd == 0 else f'{nd} differing elements'}", flush=True)
        bad += nd != 0

    K.ensure_workspace(dev)
    for (B, H, Cin, Cout) in [(32, 256, 128, 128), (64, 64, 256, 256), (64, 32, 512, 512), (16, 128, 64, 192)]:
        conv = Conv2d(Cin, Cout, 3, 1, 1).to(dev)
        w, wt, bias = conv.packed(torch.float32)
        x = torch.randn(B, H, H, Cin, device=dev)
        r = torch.randn(B, H, H, Cout, device=dev)
        dy = torch.randn(B, H, H, Cout, device=dev)
        m = torch.randn(B, H, H, Cin, device=dev)
        d = conv._desc(x)
        assert K._x3_halo(d, x, False) and K._x3_halo(d, x, True)
        check(f"x3 halo fwd         B{B} {H}x{H} {Cin}->{Cout}", lambda: K.conv2d_fwd(d, x, w, bias))
        check(f"x3 halo fwd+res     B{B} {H}x{H} {Cin}->{Cout}", lambda: K.conv2d_fwd(d, x, w, bias, r))
        check(f"x3 halo fwd+relu    B{B} {H}x{H} {Cin}->{Cout}", lambda: K.conv2d_fwd(d, x, w, bias, act=K.ACT_RELU))
        check(f"x3 halo dgrad       B{B} {H}x{H} {Cin}->{Cout}", lambda: K.conv2d_dgrad(d, dy, wt))
        check(f"x3 halo dgrad+gate  B{B} {H}x{H} {Cin}->{Cout}", lambda: K.conv2d_dgrad(d, dy, wt, m, K.ACT_LRELU))
        del x, r, dy, m
    print(f"[{rank}] " + ("FAILED" if bad else "all deterministic"), flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        sys.exit(1 if worker(int(sys.argv[2])) else 0)
    import subprocess
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r)]) for r in range(nproc)]
    rc = [p.wait() for p in procs]
    print("exit codes", rc)
    sys.exit(1 if any(rc) else 0)
