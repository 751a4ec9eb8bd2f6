#!/usr/bin/env python3
"""Is the in-step slowdown of the halo forward (4 - 9 % against tools/debug/halo_data_probe.py on the same box) the OPERANDS?  Capture the
arguments of the first `forward + statistics` 128 -> 128 @256^2 launch of a headline training step, then time that very call (a) on the
captured activations, (b) on swish(N(0,1)) of the same shape, (c) on zeros -- same kernel, same weights, same launch, back to back.

    python tools/debug/r5_halo_real_data.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from dynamicvectorquantization_amd import _lib, kernels as K, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer, reference_learning_rate

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
BS = 64
model = instantiate_from_config(bench.full_config("full", BS)).to(dev)
model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, 1, BS)
model.training_steps, model.steps_per_epoch = 100000, 1000
model.train()
tr = Trainer(model, max_steps=8, use_graph=False)
batches = [{"image": torch.from_numpy(synth.half_flat_images(BS, 256, seed=1234 + 1000 * i)).to(dev)} for i in range(2)]
for i in range(3):
    tr.train_step(batches[i % 2], i)
cap = {}
orig = K.conv2d_fwd


def spy(d, x, w, bias, residual=None, gn_ss=None, out_stats=None, out_groups=0, act=K.ACT_NONE):
    key = ("res" if residual is not None else "") + ("gn" if gn_ss is not None else "") + ("st" if out_stats is not None else "")
    if (d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.stride, d.upsample) == (BS, 256, 256, 128, 128, 3, 1, 0) and act == K.ACT_NONE and key not in cap:
        cap[key] = dict(d=d, x=x.clone(), w=w.clone(), bias=None if bias is None else bias.clone(),
                        residual=None if residual is None else residual.clone(), gn_ss=None if gn_ss is None else gn_ss.clone(),
                        groups=out_groups, stats=out_stats is not None)
    return orig(d, x, w, bias, residual, gn_ss=gn_ss, out_stats=out_stats, out_groups=out_groups, act=act)


K.conv2d_fwd = spy
import dynamicvectorquantization_amd.layers as L
tr.train_step(batches[1], 3)
K.conv2d_fwd = orig
torch.cuda.synchronize()
del tr, model


def timed(c, x, reps=20):
    def call():
        st = K.zeros_small((c["d"].N, 32, 2), torch.float64, dev) if c["stats"] else None
        return orig(c["d"], x, c["w"], c["bias"], c["residual"], gn_ss=c["gn_ss"], out_stats=st, out_groups=c["groups"] if c["stats"] else 0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        call()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def timed_after_producer(c, x, reps=20, fresh=True):
    """the same call, but -- like in the step -- its input has JUST been written by an HBM-bound kernel (a copy of x) into a buffer the
    allocator hands out for the occasion; only the convolution is bracketed by the events"""
    tot = 0.0
    keep = []
    for r in range(reps + 3):
        xi = x.clone() if fresh else x
        st = K.zeros_small((c["d"].N, 32, 2), torch.float64, dev) if c["stats"] else None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = orig(c["d"], xi, c["w"], c["bias"], c["residual"], gn_ss=c["gn_ss"], out_stats=st, out_groups=c["groups"] if c["stats"] else 0)
        e.record()
        keep.append((s, e))
        del xi, y
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in keep[3:]) / reps


print("variant            captured activations   swish(N(0,1))   N(0,1)   zeros      [ms per launch, 64 x 256^2 x 128 -> 128]")
for key in sorted(cap):
    c = cap[key]
    g = torch.Generator(device=dev).manual_seed(5)
    rn = torch.randn(c["x"].shape, device=dev, generator=g)
    xs = (rn * torch.sigmoid(rn)).to(torch.bfloat16)
    xn = rn.to(torch.bfloat16)
    xz = torch.zeros_like(c["x"])
    xf = c["x"].float()
    row = [timed(c, c["x"]), timed(c, xs), timed(c, xn), timed(c, xz), timed(c, c["x"])]
    print(f"fwd+{key or 'plain':10s}  {row[0]:.4f} (again {row[4]:.4f})      {row[1]:.4f}        {row[2]:.4f}   {row[3]:.4f}    "
          f"input mean {float(xf.mean()):+.3f} std {float(xf.std()):.3f} zero fraction {float((xf == 0).float().mean()):.3f}")
    print(f"      the same call right after an HBM-bound producer wrote its input (events around the convolution only): "
          f"{timed_after_producer(c, c['x']):.4f} ms; events per launch without the producer: {timed_after_producer(c, c['x'], fresh=False):.4f} ms")
