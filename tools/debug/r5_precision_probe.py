#!/usr/bin/env python3
"""Round-5 measurements on the headline model after a short training run:

 (a) the rows / codebook the code search sees in training: norms, how many rows the fp64 re-rank takes under the shipped error bound
     and under alternatives (per-code norms, mean-centred operands) -- evaluated in fp64 on the device, test code only;
     a sample of the operands is saved for offline work (gpurun_out/r5_vq_operands.npz);
 (b) bf16 vs fp32 layer by layer: relative error of every saved layer input of an eval forward through both instantiations;
 (c) the mixed-precision experiment: trunk in bf16, everything from encoder level L on (<= 32^2 maps, heads, conv_out_*, quant_conv,
     the search rows) in fp32 -- code mismatch rate / reconstruction error against the all-fp32 path, and what the fp32 tail costs.

    python tools/debug/r5_precision_probe.py [steps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from dynamicvectorquantization_amd import _lib, kernels as K, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.layers import Tape, _child, norm_swish_conv
from dynamicvectorquantization_amd.trainer import Trainer

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
BS = 64
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
VQ_ONLY = len(sys.argv) > 2 and sys.argv[2] == "vq_only"
WARMUP0 = os.environ.get("DVQ_PROBE_WARMUP0", "0") == "1"     # no LR warm-up (the round-4 bench configuration): the codebook moves at once
REPO = bench.REPO
out = {}


def say(*a):
    print(*a, flush=True)


model = instantiate_from_config(bench.full_config("full", BS)).to(dev)
from dynamicvectorquantization_amd.trainer import reference_learning_rate
model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, 1, BS)
model.training_steps, model.steps_per_epoch = 100000, 1000
if WARMUP0:
    model.warmup_epochs = 0
model.train()
tr = Trainer(model, max_steps=STEPS)
batches = [{"image": torch.from_numpy(synth.half_flat_images(BS, 256, seed=1234 + 1000 * i)).to(dev)} for i in range(2)]
t0 = time.time()
for i in range(STEPS):
    tr.train_step(batches[i % 2], i)
torch.cuda.synchronize()
say(f"{STEPS} train steps in {time.time() - t0:.1f} s")
tr.drop_graph()

# ------------------------------------------------------------------------------------------------------------------------------
# (a) search operands in training
seen = {}
orig = model.quantize.fwd


def spy(h, mask, tape):
    if "x" not in seen:
        seen["x"] = h.reshape(-1, h.shape[-1]).detach().clone()
        seen["cb"] = model.quantize.codebook._codebook().clone()
    return orig(h, mask, tape)


model.quantize.fwd = spy
was = tr.use_graph
tr.use_graph = False
tr.train_step(batches[0], STEPS)
tr.use_graph = was
model.quantize.fwd = orig
torch.cuda.synchronize()
x, cb = seen["x"], seen["cb"]
n, d = x.shape
k = cb.shape[0]
x64, cb64 = x.double(), cb.double()
xn, en = x64.norm(dim=1), cb64.norm(dim=1)
q = torch.tensor([0.0, 0.01, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64, device=dev)
vq = {"rows": [n, d], "codes": k, "row_dtype": str(x.dtype),
      "row_norm_quantiles": [round(float(v), 4) for v in torch.quantile(xn, q)],
      "code_norm_quantiles": [round(float(v), 5) for v in torch.quantile(en, q)],
      "code_norm_sorted_top8": [round(float(v), 4) for v in en.sort(descending=True).values[:8]],
      "row_mean_norm": round(float(x64.mean(0).norm()), 4), "code_mean_norm": round(float(cb64.mean(0).norm()), 4),
      "row_spread_rms": round(float((x64 - x64.mean(0)).norm(dim=1).pow(2).mean().sqrt()), 4),
      "cluster_size_ema_nonzero": int((model.quantize.codebook.cluster_size_ema > 1e-3).sum())}
prep = K.vq_prepare(cb)
idx, flagged = K.vq_argmin(x, cb, prep, impl=2, return_flagged=True)
vq["kernel_flagged"] = [int(v) for v in flagged.cpu()]
# exact scores in fp64 (test code): s_k = |e_k|^2 - 2 x.e_k
S = en.pow(2)[None, :] - 2.0 * (x64 @ cb64.T)
best, bi = S.min(dim=1)
vq["kernel_idx_exact"] = bool(torch.equal(idx, bi))
gap = S - best[:, None]
usage = torch.bincount(bi, minlength=k)
vq["codes_used_by_this_batch"] = int((usage > 0).sum())
vq["norm_of_used_codes_quantiles"] = [round(float(v), 4) for v in torch.quantile(en[usage > 0], q)]
D = d
coefA = 4.0 * (3.0 * 3.8147e-6 + D * 1.1921e-7)
coefB = 8.0 * 5.9605e-8
emax = en.max()
tau_g = coefA * xn * emax + coefB * (emax * emax + 2.0 * xn * emax + xn * xn)
cnt = lambda m: int(((m.sum(dim=1)) >= 2).sum())             # rows with a second code inside the bound (the best always is)
vq["rows_ambiguous_global_emax"] = cnt(gap <= tau_g[:, None])
eb = en[bi]
tau_pc = 0.5 * coefA * xn[:, None] * (en[None, :] + eb[:, None]) + coefB * (emax * emax + 2.0 * xn[:, None] * emax + xn[:, None] ** 2)
vq["rows_ambiguous_per_code_norm"] = cnt(gap <= tau_pc)
# per residue class maximum (k mod 32) + the best code's own norm
cls = torch.arange(k, device=dev) % 32
emc = torch.zeros(32, dtype=torch.float64, device=dev).scatter_reduce(0, cls, en, "amax")
tau_cl = 0.5 * coefA * xn[:, None] * (emc[cls][None, :] + eb[:, None]) + coefB * (emax * emax + 2.0 * xn[:, None] * emax + xn[:, None] ** 2)
vq["rows_ambiguous_class_max_norm"] = cnt(gap <= tau_cl)
for tag, mu in (("codebook_mean", cb64.mean(0)), ("row_mean", x64.mean(0)),
                ("used_code_mean", (cb64 * usage[:, None]).sum(0) / usage.sum())):
    xc, ec = (x64 - mu).norm(dim=1), (cb64 - mu).norm(dim=1)
    ecm = ec.max()
    t_glob = coefA * xc * ecm + coefB * (ecm * ecm + 2.0 * xc * ecm + xc * xc)
    t_pc = 0.5 * coefA * xc[:, None] * (ec[None, :] + ec[bi][:, None]) + coefB * (ecm * ecm + 2.0 * xc[:, None] * ecm + xc[:, None] ** 2)
    vq[f"centred_{tag}"] = {"row_norm_median": round(float(xc.median()), 4), "code_norm_max": round(float(ecm), 4),
                            "rows_ambiguous_global": cnt(gap <= t_glob[:, None]), "rows_ambiguous_per_code": cnt(gap <= t_pc)}
g2 = gap.clone()
g2.scatter_(1, bi[:, None], float("inf"))
top2 = g2.min(dim=1).values
vq["top2_gap_quantiles"] = [float(v) for v in torch.quantile(top2, q)]
vq["tau_global_median"] = float(tau_g.median())
out["vq_in_training"] = vq
say(json.dumps({"vq_in_training": vq}))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
sel = torch.randperm(n, device=dev)[:8192]
np.savez_compressed(os.path.join(REPO, "gpurun_out", "r5_vq_operands_warmup0.npz" if WARMUP0 else "r5_vq_operands.npz"), x=x[sel].float().cpu().numpy().astype(np.float32),
                    x_is_bf16=np.array(x.dtype == torch.bfloat16), cb=cb.cpu().numpy(), usage=usage.cpu().numpy())
del S, gap, g2, tau_pc, tau_cl, x64, cb64
if VQ_ONLY:
    sys.exit(0)

# ------------------------------------------------------------------------------------------------------------------------------
# (b) layer-by-layer bf16 vs fp32 (eval forward, trained weights)
model.eval()
xin = torch.from_numpy(synth.half_flat_images(BS, 256, seed=4321)).to(dev)


def walk(t, prefix=""):
    for kk, v in t.s.items():
        if kk == "x" and torch.is_tensor(v):
            yield prefix, v
    for name, c in t.c.items():
        yield from walk(c, prefix + "/" + name)


tapes, outs = {}, {}
with torch.no_grad():
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        with rt.compute_dtype_ctx(dt):
            tp = Tape()
            model.ae_fwd(xin[:16], tp)                     # layer inputs: 16 images (both tapes stay resident)
            tapes[tag] = dict(walk(tp))
            del tp
            o = model.ae_fwd(xin, None)                    # codes / reconstructions: the whole batch, nothing saved
            outs[tag] = {kk: (v.float().clone() if torch.is_tensor(v) and v.is_floating_point() else v) for kk, v in o.items()
                         if kk in ("rec", "codes", "grain", "quant")}
layers = []
for name, vb in tapes["bf16"].items():
    vf = tapes["fp32"].get(name)
    if vf is None or vf.numel() != vb.numel():
        continue
    a, b = vb.float().reshape(-1), vf.float().reshape(-1)
    layers.append((name, round(float((a - b).norm() / b.norm().clamp_min(1e-30)), 5)))
out["layer_input_rel_err_bf16_vs_fp32"] = layers
say("layer inputs, ||bf16 - fp32|| / ||fp32||:")
for name, e in layers:
    say(f"  {e:9.5f}  {name}")
del tapes


def code_cells_mismatch(ca, cf, grain):
    rep = grain.repeat_interleave(ca.shape[1] // grain.shape[1], 1).repeat_interleave(ca.shape[2] // grain.shape[2], 2).bool()
    diff = ca != cf
    cells = int(rep.sum()) + int((~rep).sum()) // 4
    return int((diff & rep).sum()) + int((diff & ~rep).sum()) // 4, cells


m_bf, cells = code_cells_mismatch(outs["bf16"]["codes"], outs["fp32"]["codes"], outs["fp32"]["grain"])
base = {"code_mismatches": m_bf, "cells": cells, "rate": round(m_bf / cells, 6),
        "recon_rel_err": round(float((outs["bf16"]["rec"] - outs["fp32"]["rec"]).norm() / outs["fp32"]["rec"].norm()), 6)}
say("all-bf16 vs all-fp32:", base)


# ------------------------------------------------------------------------------------------------------------------------------
# (c) mixed precision: encoder from level `lvl_switch` on in fp32 (forward only -- codes and reconstructions are forward quantities)
def encoder_fwd_mixed(enc, x_img, grain, lvl_switch):
    """_GrainEncoder.fwd for the fixed-entropy dual encoder with a precision switch at the entry of level `lvl_switch`"""
    cd = torch.bfloat16
    x_ = K.nchw_to_nhwc_pad(x_img, K.vec(cd) * -(-enc.in_channels // K.vec(cd)), cd)
    h = enc.conv_in.fwd(x_, None)
    taps = {}
    s = len(enc.HEADS)
    for i_level in range(enc.num_resolutions):
        if i_level == lvl_switch:
            h = K.cast(h, torch.float32)
        lvl = enc.down[i_level]
        for i_block in range(enc.num_res_blocks):
            h = lvl.block[i_block].fwd(h, None)
            if len(lvl.attn) > 0:
                h = lvl.attn[i_block].fwd(h, None)
        kk = enc.num_resolutions - 1 - i_level
        if 0 < kk < s:
            taps[kk] = h
        if i_level != enc.num_resolutions - 1:
            h = lvl.downsample.fwd(h, None)
    taps[0] = h
    heads = []
    for kk, name in enumerate(enc.HEADS):
        mid = getattr(enc, f"mid_{name}")
        t = taps[kk]
        if lvl_switch == 99 and t.dtype != torch.float32:      # heads only
            t = K.cast(t, torch.float32)
        t = mid.block_1.fwd(t, None)
        t = mid.attn_1.fwd(t, None)
        t = mid.block_2.fwd(t, None)
        heads.append(norm_swish_conv(getattr(enc, f"norm_out_{name}"), getattr(enc, f"conv_out_{name}"), t, None, "n", "c"))
    if any(h_.dtype == torch.float32 for h_ in heads):           # a switch below the fine tap leaves the fine head in bf16
        heads = [h_ if h_.dtype == torch.float32 else K.cast(h_, torch.float32) for h_ in heads]
    merged, mask = K.dual_merge(heads[1], heads[0], grain)
    return merged, mask


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        r = fn()
    e.record()
    torch.cuda.synchronize()
    return r, s.elapsed_time(e) / reps


mixed = []
with torch.no_grad():
    ent, gate = K.patch_entropy_gate(xin, model.entropy_patch_size, model._threshold())
    grain = gate[..., 1].contiguous()
    for lvl in (5, 99, 4, 3, 2, 1, 0):        # 5 = no switch (all bf16 through this code path); 99 = heads (mid_*, conv_out_*) + quant_conv only
        def enc_part():
            return encoder_fwd_mixed(model.encoder, xin, grain, lvl)
        (merged, mask), ms_enc = timed(enc_part)
        if lvl != 5 and merged.dtype != torch.float32:
            merged = K.cast(merged, torch.float32)
        h = model.quant_conv.fwd(merged, None)
        cbk = model.quantize.codebook
        idxm = cbk.find_nearest_embedding(h.view(-1, h.shape[-1])).view(h.shape[:3])
        xq, _ = K.vq_gather_loss(h.view(-1, h.shape[-1]), cbk._codebook(), idxm.view(-1), mask.reshape(-1))
        xq = xq.view(h.shape)
        z = model.post_quant_conv.fwd(K.cast(xq, torch.bfloat16) if xq.dtype != torch.bfloat16 else xq, None)
        rec = K.nhwc_pad_to_nchw(model.decoder.fwd(z, None), model.decoder.out_ch).float()
        mm, cells = code_cells_mismatch(idxm, outs["fp32"]["codes"], outs["fp32"]["grain"])
        row = {"fp32_from_level": lvl, "encoder_fwd_ms": round(ms_enc, 2), "code_mismatches": mm, "rate": round(mm / cells, 6),
               "recon_rel_err": round(float((rec - outs["fp32"]["rec"]).norm() / outs["fp32"]["rec"].norm()), 6)}
        mixed.append(row)
        say("mixed:", row)
out["all_bf16_vs_fp32"] = base
out["mixed_precision_forward"] = mixed
with open(os.path.join(REPO, "gpurun_out", "r5_precision_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
say("written gpurun_out/r5_precision_probe.json")
