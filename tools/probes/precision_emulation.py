#!/usr/bin/env python3
"""Bounded experiment: what would an 11-bit-mantissa (fp16) STORAGE format buy over bf16 for the autoencoder forward?
Emulated on the host with the oracle (oracle/dqvae.py), no kernels: every tensor a kernel would store -- convolution outputs, the
GroupNorm+swish result, residual sums, attention operands -- and every weight is rounded to the storage type, products and sums stay
fp32 (what the MFMA does).  Reported against the fp32 oracle on the shipped geometry (ch 128, codebook 1024 x 256), 256 x 256 images:
relative L2 error of the reconstruction, of the decoder alone on identical codes, and the fraction of code indices that differ.
    python tools/probes/precision_emulation.py [--bs 4] [--size 256]
TEST / ANALYSIS INFRASTRUCTURE: imports oracle/, never imported by the product."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dynamicvectorquantization_amd import synth  # noqa: E402
from oracle import dqvae as odq  # noqa: E402
from oracle import entropy as oent  # noqa: E402
from oracle import vq as ovq  # noqa: E402

MODE = {"dtype": None}
_conv, _gn, _swish = odq.conv, odq.group_norm, odq.swish


def rnd(t):
    return t if MODE["dtype"] is None else t.to(MODE["dtype"]).float()


def conv(sd, prefix, x, stride=1, padding=0):
    w = rnd(sd[prefix + ".weight"])
    return rnd(torch.nn.functional.conv2d(rnd(x), w, sd.get(prefix + ".bias"), stride=stride, padding=padding))


def swish(x):
    return rnd(_swish(x))           # GroupNorm + swish are one kernel: one rounding, after the activation


odq.conv, odq.swish = conv, swish


def run(sd, x, thr, codes_override=None):
    ent = oent.patch_entropy(x.numpy())
    enc = odq.encoder_dual(sd, x, ent, thr)
    h = odq.conv(sd, "quant_conv", enc["h_dual"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d).numpy()
    cb = sd["quantize.codebook.weight"][:-1].numpy()
    codes = ovq.argmin_exact(flat, cb) if codes_override is None else codes_override
    xq = torch.from_numpy(cb[codes].astype(np.float32)).reshape(b, hh, ww, d).permute(0, 3, 1, 2)
    z = odq.conv(sd, "post_quant_conv", xq)
    return codes, odq.decoder(sd, z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    from dynamicvectorquantization_amd.config import instantiate_from_config, stage1_config
    torch.manual_seed(0)
    cfg = stage1_config(objective="none")
    if a.size != 256:
        g = dict(synth.DQVAE_GEOM["c1"])
        g.update(resolution=a.size, latent=a.size // 8)
        cfg = stage1_config(objective="none", geometry=g)
    model = instantiate_from_config(cfg.model)              # reference-identical initialisation (codebook U(+-1/K): near ties everywhere)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    thr = oent.threshold_from_table(os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    x = torch.from_numpy(synth.half_flat_images(a.bs, a.size, seed=1234))
    res = {}
    with torch.no_grad():
        for cbname in ("refinit", "spread"):
            if cbname == "spread":
                k, zc = sd["quantize.codebook.weight"].shape[0] - 1, sd["quantize.codebook.weight"].shape[1]
                sd["quantize.codebook.weight"] = torch.from_numpy(synth.det_param("quantize.codebook.weight.spread", (k + 1, zc)) * np.sqrt(zc) * 1.2)
            MODE["dtype"] = None
            c32, r32 = run(sd, x, thr)
            for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
                MODE["dtype"] = dt
                c, r = run(sd, x, thr)
                _, rdec = run(sd, x, thr, codes_override=c32)
                l2 = lambda u, v: float((u - v).norm() / v.norm())      # noqa: E731
                res[(cbname, name)] = (float((c != c32).mean()), l2(r, r32), l2(rdec, r32), float(r.abs().max()))
                print(f"codebook {cbname:8s} storage {name}: codes differing {res[(cbname, name)][0]:.2e}   recon rel L2 {res[(cbname, name)][1]:.2e}   "
                      f"decoder-only (same codes) {res[(cbname, name)][2]:.2e}   max |act| seen in recon {res[(cbname, name)][3]:.2f}", flush=True)


if __name__ == "__main__":
    main()
