"""Oracle: the DQ-VAE CNN blocks, dual-grain encoder, positional decoder and whole-model
forward, as pure functions of a state_dict (torch-CPU fp32).

Follows (behaviour, not code), all under /root/reference:
  * nonlinearity / Normalize / Upsample / Downsample / ResnetBlock / AttnBlock
        modules/diffusionmodules/model.py:29-192
  * DualGrainEncoder.forward            modules/dynamic_modules/EncoderDual.py:89-156
  * Decoder.forward + position biases   modules/dynamic_modules/DecoderPositional.py:27-39,109-146
                                        modules/dynamic_modules/fourier_embedding.py:45-55
  * DualGrainVQModel.encode/decode      models/stage1_dynamic/dqvae_dual_entropy.py:124-144

A "state_dict" here is {name: torch.Tensor (cpu, fp32)} with the reference's key names, so the
same dict drives the reference module (in tools/gen_golden.py), this oracle and the HIP modules.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import entropy as oent
from . import vq as ovq


def swish(x):
    return x * torch.sigmoid(x)


def group_norm(sd, prefix, x, groups=32, eps=1e-6):
    return F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def conv(sd, prefix, x, stride=1, padding=0):
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)


def resnet_block(sd, prefix, x, drop_mask=None):
    """drop_mask (tests of dropout > 0, model.py:127): multiplier [B,C,H,W] (0 or 1 / (1 - p)) applied to the second activation"""
    cin = sd[prefix + ".conv1.weight"].shape[1]
    cout = sd[prefix + ".conv1.weight"].shape[0]
    h = conv(sd, prefix + ".conv1", swish(group_norm(sd, prefix + ".norm1", x)), padding=1)
    a = swish(group_norm(sd, prefix + ".norm2", h))
    if drop_mask is not None:
        a = a * drop_mask
    h = conv(sd, prefix + ".conv2", a, padding=1)
    if cin != cout:
        if prefix + ".conv_shortcut.weight" in sd:       # conv_shortcut=True (model.py:103-108,131-135): a 3x3 instead of the 1x1
            x = conv(sd, prefix + ".conv_shortcut", x, padding=1)
        else:
            x = conv(sd, prefix + ".nin_shortcut", x)
    return x + h


def attn_block(sd, prefix, x):
    b, c, hh, ww = x.shape
    n = hh * ww
    hn = group_norm(sd, prefix + ".norm", x)
    q = conv(sd, prefix + ".q", hn).reshape(b, c, n)
    k = conv(sd, prefix + ".k", hn).reshape(b, c, n)
    v = conv(sd, prefix + ".v", hn).reshape(b, c, n)
    scores = torch.einsum("bci,bcj->bij", q, k) * (int(c) ** (-0.5))
    p = torch.softmax(scores, dim=2)
    o = torch.einsum("bcj,bij->bci", v, p).reshape(b, c, hh, ww)
    return x + conv(sd, prefix + ".proj_out", o)


def downsample(sd, prefix, x):
    if prefix + ".conv.weight" not in sd:            # with_conv=False (model.py:73-74): 2 x 2 average pooling
        return F.avg_pool2d(x, kernel_size=2, stride=2)
    return conv(sd, prefix + ".conv", F.pad(x, (0, 1, 0, 1)), stride=2)


def upsample(sd, prefix, x):
    x = x.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)
    if prefix + ".conv.weight" not in sd:            # with_conv=False (model.py:49-53): nearest x2 only
        return x
    return conv(sd, prefix + ".conv", x, padding=1)


def _count(sd, pattern):
    i = 0
    while any(k.startswith(pattern.format(i)) for k in sd):
        i += 1
    return i


def encoder_dual(sd, x, x_entropy, threshold, prefix="encoder"):
    """Entropy-routed dual-grain encoder.  Returns dict(h_dual, indices, codebook_mask, gate,
    h_coarse, h_fine)."""
    p = prefix
    n_levels = _count(sd, p + ".down.{}.")
    h = conv(sd, p + ".conv_in", x, padding=1)
    h_fine = None
    for lvl in range(n_levels):
        n_blocks = _count(sd, p + f".down.{lvl}.block." + "{}.")
        has_attn = any(k.startswith(p + f".down.{lvl}.attn.0.") for k in sd)
        for blk in range(n_blocks):
            h = resnet_block(sd, p + f".down.{lvl}.block.{blk}", h)
            if has_attn:
                h = attn_block(sd, p + f".down.{lvl}.attn.{blk}", h)
        if lvl == n_levels - 2:
            h_fine = h
        if lvl != n_levels - 1:
            h = downsample(sd, p + f".down.{lvl}.downsample", h)
    hc = resnet_block(sd, p + ".mid_coarse.block_1", h)
    hc = attn_block(sd, p + ".mid_coarse.attn_1", hc)
    hc = resnet_block(sd, p + ".mid_coarse.block_2", hc)
    hc = conv(sd, p + ".conv_out_coarse", swish(group_norm(sd, p + ".norm_out_coarse", hc)), padding=1)
    hf = resnet_block(sd, p + ".mid_fine.block_1", h_fine)
    hf = attn_block(sd, p + ".mid_fine.attn_1", hf)
    hf = resnet_block(sd, p + ".mid_fine.block_2", hf)
    hf = conv(sd, p + ".conv_out_fine", swish(group_norm(sd, p + ".norm_out_fine", hf)), padding=1)

    gate = torch.as_tensor(oent.entropy_gate(np.asarray(x_entropy), threshold))     # [B,h,w,2]
    gate = gate.permute(0, 3, 1, 2)
    indices = gate.argmax(dim=1)                                                    # 1 = fine
    up = lambda t: t.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)
    idx_rep = up(indices).unsqueeze(1)
    h_dual = torch.where(idx_rep == 0, up(hc), hf)
    mask = torch.where(idx_rep == 0, torch.tensor(0.25), torch.tensor(1.0)).to(hf.dtype)
    return {"h_dual": h_dual, "indices": indices, "codebook_mask": mask, "gate": gate,
            "h_coarse": hc, "h_fine": hf}


def position_bias(sd, latent, prefix="decoder"):
    """fourier+learned additive bias [1,C,latent,latent] (DecoderPositional.py:27-39,
    fourier_embedding.py:7-15,45-55)."""
    lin = torch.linspace(-1, 1, latent)
    xs = lin.view(1, 1, 1, -1).repeat(1, 1, latent, 1)
    ys = lin.view(1, 1, -1, 1).repeat(1, 1, 1, latent)
    coord = torch.cat([xs, ys], dim=1).to(sd[prefix + ".position_bias_fourier.lff.ffm.conv.weight"].dtype)
    four = torch.sin(conv(sd, prefix + ".position_bias_fourier.lff.ffm.conv", coord))
    col = sd[prefix + ".position_bias_learned.col_embed.weight"][:latent]          # [w, C]
    row = sd[prefix + ".position_bias_learned.row_embed.weight"][:latent]          # [h, C]
    learned = (col.unsqueeze(0) + row.unsqueeze(1)).permute(2, 0, 1).unsqueeze(0)   # [1,C,h,w]
    return four, learned


def decoder(sd, z, prefix="decoder", give_pre_end=False):
    """position_type is read off the state_dict keys: `position_bias_fourier` + `position_bias_learned` = "fourier+learned" (the shipped
    configs); a lone `position_bias.lff...` = "fourier"; a lone `position_bias.row_embed` = "learned" / "learned-relative", for which the
    reference's forward (DecoderPositional.py:112-123) has no branch: the parameter exists and NOTHING is added."""
    p = prefix
    if p + ".position_bias_fourier.lff.ffm.conv.weight" in sd:
        four, learned = position_bias(sd, z.shape[-1], p)
        h = (z + four) + learned
    elif p + ".position_bias.lff.ffm.conv.weight" in sd:
        lin = torch.linspace(-1, 1, z.shape[-1])
        coord = torch.cat([lin.view(1, 1, 1, -1).repeat(1, 1, z.shape[-1], 1), lin.view(1, 1, -1, 1).repeat(1, 1, 1, z.shape[-1])], dim=1)
        h = z + torch.sin(conv(sd, p + ".position_bias.lff.ffm.conv", coord))
    else:
        h = z
    h = conv(sd, p + ".conv_in", h, padding=1)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    n_levels = _count(sd, p + ".up.{}.")
    for lvl in reversed(range(n_levels)):
        n_blocks = _count(sd, p + f".up.{lvl}.block." + "{}.")
        has_attn = any(k.startswith(p + f".up.{lvl}.attn.0.") for k in sd)
        for blk in range(n_blocks):
            h = resnet_block(sd, p + f".up.{lvl}.block.{blk}", h)
            if has_attn:
                h = attn_block(sd, p + f".up.{lvl}.attn.{blk}", h)
        if lvl != 0:
            h = upsample(sd, p + f".up.{lvl}.upsample", h)
    if give_pre_end:                                       # DecoderPositional.py:139-140
        return h
    return conv(sd, p + ".conv_out", swish(group_norm(sd, p + ".norm_out", h)), padding=1)


def dqvae_forward(sd, x, threshold, patch=16, beta=0.25):
    """Eval-mode DualGrainVQModel.forward.  x: torch [B,3,H,W].  Returns dict of torch/np outputs."""
    ent = oent.patch_entropy(x.numpy(), patch=patch)
    enc = encoder_dual(sd, x, ent, threshold)
    h = conv(sd, "quant_conv", enc["h_dual"])
    xq, qloss, codes = ovq.vq_forward(h.numpy(), sd["quantize.codebook.weight"].numpy(),
                                      enc["codebook_mask"].numpy(), beta=beta)
    z = conv(sd, "post_quant_conv", torch.as_tensor(xq))
    rec = decoder(sd, z)
    return {"entropy": ent, "h_dual": enc["h_dual"], "grain_indices": enc["indices"], "gate": enc["gate"],
            "codebook_mask": enc["codebook_mask"], "h_quant_in": h, "codes": codes, "x_q": xq,
            "qloss": qloss, "rec": rec}
