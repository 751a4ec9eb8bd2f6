#!/usr/bin/env python3
"""Micro-probe of the dominant conv shape (128->128 3x3 @256x256, B=64, bf16): fwd / dgrad / wgrad timings
with HIP events; run under `rocprofv3 --pmc ...` to attribute HBM traffic per dispatch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d, Tape

def main():
    B = int(os.environ.get("PROBE_B", 64)); H = int(os.environ.get("PROBE_H", 256)); C = int(os.environ.get("PROBE_C", 128))
    reps = int(os.environ.get("PROBE_REPS", 5))
    dev = torch.device("cuda:0")
    rt.set_compute_dtype(torch.bfloat16)
    rt.set_impl(int(os.environ.get("DVQ_IMPL", 0)))
    torch.manual_seed(0)
    conv = Conv2d(C, C, 3, 1, 1).to(dev)
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    res = {}
    flops = 2 * B * H * H * C * C * 9
    def timeit(name, fn):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        res[name] = {"ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1)}
    tape = Tape()
    w, wt, bias = conv.packed(torch.bfloat16)
    d = conv._desc(x)
    timeit("fwd", lambda: K.conv2d_fwd(d, x, w, bias, None))
    timeit("fwd_res", lambda: K.conv2d_fwd(d, x, w, bias, dy))
    timeit("dgrad", lambda: K.conv2d_dgrad(d, dy, wt))
    timeit("wgrad", lambda: K.conv2d_wgrad(d, x, dy, None))
    print(json.dumps(res))

if __name__ == "__main__":
    main()
