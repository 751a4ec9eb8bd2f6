#!/usr/bin/env python3
"""Per-shape conv timings (bf16): every distinct 3x3 conv shape of the DQ-VAE step (AE, VGG16 of LPIPS) and the 4x4 PatchGAN
convs: fwd / dgrad / wgrad ms and TFLOP/s, HIP events.  PROBE_SET=vgg|ae|disc|all"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d

SETS = {
    # (N, H, Cin, Cout, k, stride, pad, modes)
    "vgg": [(128, 256, 3, 64, 3, 1, 1, "fd"), (128, 256, 64, 64, 3, 1, 1, "fd"), (128, 128, 64, 128, 3, 1, 1, "fd"),
            (128, 128, 128, 128, 3, 1, 1, "fd"), (128, 64, 128, 256, 3, 1, 1, "fd"), (128, 64, 256, 256, 3, 1, 1, "fd"),
            (128, 32, 256, 512, 3, 1, 1, "fd"), (128, 32, 512, 512, 3, 1, 1, "fd"), (128, 16, 512, 512, 3, 1, 1, "fd")],
    "ae": [(64, 256, 128, 128, 3, 1, 1, "fdw"), (64, 128, 128, 128, 3, 1, 1, "fdw"), (64, 64, 128, 256, 3, 1, 1, "fdw"),
           (64, 64, 256, 256, 3, 1, 1, "fdw"), (64, 32, 256, 256, 3, 1, 1, "fdw"), (64, 32, 256, 512, 3, 1, 1, "fdw"),
           (64, 32, 512, 512, 3, 1, 1, "fdw"), (64, 16, 512, 512, 3, 1, 1, "fdw"), (64, 256, 128, 3, 3, 1, 1, "fdw"),
           (64, 256, 3, 128, 3, 1, 1, "fw")],
    "disc": [(64, 256, 3, 64, 4, 2, 1, "fdw"), (64, 128, 64, 128, 4, 2, 1, "fdw"), (64, 64, 128, 256, 4, 2, 1, "fdw"),
             (64, 32, 256, 512, 4, 1, 1, "fdw"), (64, 31, 512, 1, 4, 1, 1, "fdw")],
}


def main():
    which = os.environ.get("PROBE_SET", "all")
    reps = int(os.environ.get("PROBE_REPS", 5))
    dev = torch.device("cuda:0")
    rt.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    shapes = sum((v for k, v in SETS.items() if which in (k, "all")), [])
    for (n, h, ci, co, k, s, p, modes) in shapes:
        conv = Conv2d(ci, co, k, s, p).to(dev)
        cip, cop = conv._padded(torch.bfloat16)
        x = torch.randn(n, h, h, cip, device=dev).to(torch.bfloat16)
        d = conv._desc(x)
        dy = torch.randn(n, d.OH, d.OW, cop, device=dev).to(torch.bfloat16)
        w, wt, bias = conv.packed(torch.bfloat16)
        gw = torch.zeros(co, k, k, ci, device=dev).permute(0, 3, 1, 2)   # OHWI storage like the trainer's flat buffers
        flops = 2.0 * n * d.OH * d.OW * ci * co * k * k
        row = {}

        def timeit(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(reps):
                fn()
            en.record()
            torch.cuda.synchronize()
            return st.elapsed_time(en) / reps
        if "f" in modes:
            row["fwd"] = timeit(lambda: K.conv2d_fwd(d, x, w, bias, act=K.ACT_RELU))
        if "d" in modes:
            row["dgrad"] = timeit(lambda: K.conv2d_dgrad(d, dy, wt, mask=x if ci % 8 == 0 else None, mask_act=K.ACT_RELU))
        if "w" in modes:
            row["wgrad"] = timeit(lambda: K.conv2d_wgrad_oihw(d, x, dy, ci, co, gw, None))
        print(f"N={n:3d} H={h:3d} {ci:3d}->{co:3d} k{k}s{s} " + "  ".join(f"{m} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s" for m, t in row.items()),
              flush=True)


if __name__ == "__main__":
    main()
