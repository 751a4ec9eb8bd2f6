"""Oracle: feature-routed (Gumbel) dual / triple grain encoders and models (torch-CPU fp32, autograd-differentiable).
TEST INFRASTRUCTURE -- never imported by the product package.

Follows (behaviour, not code), all under /root/reference:
  * DualGrainFeatureRouter / TripleGrainFeatureRouter    modules/dynamic_modules/RouterDual.py:6-43, RouterTriple.py:6-56
  * routing tail of the encoders                         EncoderDual.py:130-156, EncoderTriple.py:143-183
  * torch.nn.functional.gumbel_softmax(hard=True)        restated with INJECTED Exp(1) noise (the device RNG draw is
                                                         "parity unpinned"; goldens patch Tensor.exponential_ in the reference)
  * TripleGrainVQModel / dual_feat DualGrainVQModel      models/stage1_dynamic/dqvae_triple_feat.py:68-88, dqvae_dual_feat.py
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dqvae as odq
from . import vq as ovq

HEADS = {2: ("coarse", "fine"), 3: ("coarse", "median", "fine")}


def feature_router(sd, prefix, heads):
    """heads: [coarsest .. finest] NCHW -> logits [B,hc,wc,S]"""
    s = len(heads)
    names = HEADS[s]
    feats = []
    for lvl, (name, h) in enumerate(zip(names, heads)):
        key = f"{prefix}.feature_norm_{name}.weight"
        if key in sd:
            groups = 32
            h = F.group_norm(h, groups, sd[key], sd[f"{prefix}.feature_norm_{name}.bias"], 1e-6)
        if lvl > 0:
            h = F.avg_pool2d(h, 1 << lvl, 1 << lvl)
        feats.append(h)
    x = torch.cat(feats, dim=1).permute(0, 2, 3, 1)
    if f"{prefix}.gate.weight" in sd:
        return F.linear(x, sd[f"{prefix}.gate.weight"], sd[f"{prefix}.gate.bias"])
    act = F.relu if sd.get(f"{prefix}.gate_type") == "2layer-fc-ReLu" else F.silu      # (the activation has no parameters: the caller says)
    x = act(F.linear(x, sd[f"{prefix}.gate.0.weight"], sd[f"{prefix}.gate.0.bias"]))
    return F.linear(x, sd[f"{prefix}.gate.2.weight"], sd[f"{prefix}.gate.2.bias"])


def gumbel_hard(logits, exponential, tau=1.0):
    y_soft = ((logits - exponential.log()) / tau).softmax(dim=-1)
    index = y_soft.max(dim=-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
    return y_hard - y_soft.detach() + y_soft


def encoder_feature_routed(sd, x, n_heads, exponential=None, prefix="encoder"):
    """exponential: Exp(1) noise [B,hc,wc,S] -> training-mode Gumbel routing; None -> eval (raw logits).
    Returns dict(h, indices, codebook_mask, gate)."""
    p = prefix
    names = HEADS[n_heads]
    n_levels = odq._count(sd, p + ".down.{}.")
    h = odq.conv(sd, p + ".conv_in", x, padding=1)
    taps = {}
    for lvl in range(n_levels):
        n_blocks = odq._count(sd, p + f".down.{lvl}.block." + "{}.")
        has_attn = any(k.startswith(p + f".down.{lvl}.attn.0.") for k in sd)
        for blk in range(n_blocks):
            h = odq.resnet_block(sd, p + f".down.{lvl}.block.{blk}", h)
            if has_attn:
                h = odq.attn_block(sd, p + f".down.{lvl}.attn.{blk}", h)
        k = n_levels - 1 - lvl
        if 0 < k < n_heads:
            taps[k] = h
        if lvl != n_levels - 1:
            h = odq.downsample(sd, p + f".down.{lvl}.downsample", h)
    taps[0] = h
    heads = []
    for k, name in enumerate(names):
        t = odq.resnet_block(sd, p + f".mid_{name}.block_1", taps[k])
        t = odq.attn_block(sd, p + f".mid_{name}.attn_1", t)
        t = odq.resnet_block(sd, p + f".mid_{name}.block_2", t)
        heads.append(odq.conv(sd, p + f".conv_out_{name}", odq.swish(odq.group_norm(sd, p + f".norm_out_{name}", t)), padding=1))
    logits = feature_router(sd, p + ".router", heads)
    gate = gumbel_hard(logits, exponential) if exponential is not None else logits
    gate = gate.permute(0, 3, 1, 2)
    indices = gate.argmax(dim=1)
    f = 1 << (n_heads - 1)
    up = lambda t, r: t.repeat_interleave(r, dim=-1).repeat_interleave(r, dim=-2)
    idx_rep = up(indices, f).unsqueeze(1)
    merged = heads[-1]
    mask = torch.ones_like(idx_rep, dtype=torch.float32)
    for lvl in range(n_heads - 1):
        r = 1 << (n_heads - 1 - lvl)
        merged = torch.where(idx_rep == lvl, up(heads[lvl], r), merged)
        mask = torch.where(idx_rep == lvl, torch.tensor(1.0 / (r * r)), mask)
    if exponential is not None:
        merged = merged * up(gate.max(dim=1, keepdim=True)[0], f)
    return {"h": merged, "indices": indices, "codebook_mask": mask, "gate": gate}


def model_forward(sd, x, n_heads, exponential=None, beta=0.25):
    """feature-routed DQ-VAE forward with the exact argmin and the straight-through estimator; -> dict(rec, qloss, codes,
    indices, gate)"""
    enc = encoder_feature_routed(sd, x, n_heads, exponential)
    h = odq.conv(sd, "quant_conv", enc["h"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d)
    cb = sd["quantize.codebook.weight"][:-1].detach()
    idx = torch.from_numpy(ovq.argmin_exact(flat.detach().numpy(), cb.numpy()))
    xq = cb[idx]
    m = enc["codebook_mask"].permute(0, 2, 3, 1).reshape(-1, 1)
    qloss = beta * torch.mean((xq.detach() - flat) ** 2 * m) + torch.mean((xq - flat.detach()) ** 2 * m)
    st = flat + (xq - flat).detach()
    z = odq.conv(sd, "post_quant_conv", st.reshape(b, hh, ww, d).permute(0, 3, 1, 2))
    rec = odq.decoder(sd, z)
    return {"rec": rec, "qloss": qloss, "codes": idx.reshape(b, hh, ww), "indices": enc["indices"], "gate": enc["gate"]}
