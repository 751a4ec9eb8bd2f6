"""Oracle: one DQ-VAE autoencoder train step on CPU (torch autograd over oracle.dqvae), used by
tests and by bench.py's `cpu_baseline` leg.  TEST INFRASTRUCTURE -- never imported by the product.

Objective of the AE-only step (what the HIP trainer runs when the loss config has perceptual_weight = 0 and
disc_factor = 0): mean |x - rec| + codebook_weight * qloss, Adam(betas .5/.9) on every AE parameter.
Follows dqvae_dual_entropy.py:154-171,206-216 and vqperceptual_multidisc.py:116-153 with the LPIPS and GAN
terms switched off.
"""
from __future__ import annotations

import numpy as np
import torch

from . import dqvae as odq
from . import entropy as oent
from . import vq as ovq


def ae_loss(sd, x, threshold, beta=0.25, codebook_weight=1.0):
    """differentiable (w.r.t. the tensors in sd) AE loss; straight-through VQ with the exact argmin"""
    ent = oent.patch_entropy(x.numpy())
    enc = odq.encoder_dual(sd, x, ent, threshold)
    h = odq.conv(sd, "quant_conv", enc["h_dual"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d)
    cb = sd["quantize.codebook.weight"][:-1].detach()
    idx = torch.from_numpy(ovq.argmin_exact(flat.detach().numpy(), cb.numpy()))
    xq = cb[idx]
    m = enc["codebook_mask"].permute(0, 2, 3, 1).reshape(-1, 1)
    qloss = beta * torch.mean((xq - flat) ** 2 * m) + torch.mean((xq - flat.detach()) ** 2 * m)
    st = flat + (xq - flat).detach()
    z = odq.conv(sd, "post_quant_conv", st.reshape(b, hh, ww, d).permute(0, 3, 1, 2))
    rec = odq.decoder(sd, z)
    loss = torch.mean(torch.abs(x - rec)) + codebook_weight * qloss
    return loss, rec, idx.reshape(b, hh, ww)


def ae_forward(sd, x, threshold, beta=0.25):
    """(rec, qloss) with the straight-through estimator; shared by the full-objective step below"""
    ent = oent.patch_entropy(x.numpy())
    enc = odq.encoder_dual(sd, x, ent, threshold)
    h = odq.conv(sd, "quant_conv", enc["h_dual"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d)
    cb = sd["quantize.codebook.weight"][:-1].detach()
    idx = torch.from_numpy(ovq.argmin_exact(flat.detach().numpy(), cb.numpy()))
    xq = cb[idx]
    m = enc["codebook_mask"].permute(0, 2, 3, 1).reshape(-1, 1)
    qloss = beta * torch.mean((xq - flat) ** 2 * m) + torch.mean((xq - flat.detach()) ** 2 * m)
    st = flat + (xq - flat).detach()
    z = odq.conv(sd, "post_quant_conv", st.reshape(b, hh, ww, d).permute(0, 3, 1, 2))
    return odq.decoder(sd, z), qloss


def full_objective_steps(sd, sd_disc, sd_lpips, batches, threshold, lr=1e-4, steps=1, disc_weight_max=0.75):
    """The reference's complete two-optimizer step (dqvae_dual_entropy.py:154-171 + vqperceptual_multidisc.py:109-194):
    optimizer 0: L1 + LPIPS + adaptive hinge-GAN + codebook loss on the autoencoder; optimizer 1: a second autoencoder
    forward, then the hinge loss on the discriminator.  Adam(betas .5/.9) for both."""
    from . import losses as olo
    ae_params = [v.requires_grad_(True) for k, v in sd.items()
                 if v.dtype == torch.float32 and v.dim() > 0 and not k.startswith("quantize.")]
    d_params = [v.requires_grad_(True) for k, v in sd_disc.items() if v.dtype == torch.float32 and "running" not in k and v.dim() > 0]
    opt_ae = torch.optim.Adam(ae_params, lr=lr, betas=(0.5, 0.9))
    opt_d = torch.optim.Adam(d_params, lr=lr, betas=(0.5, 0.9))
    out = []
    for s in range(steps):
        x = batches[s % len(batches)]
        opt_ae.zero_grad(set_to_none=True)
        rec, qloss = ae_forward(sd, x, threshold)
        r = olo.generator_loss(sd_disc, sd_lpips, x, rec, qloss, sd["decoder.conv_out.weight"], disc_weight_max=disc_weight_max)
        r["loss"].backward()
        opt_ae.step()
        for p in d_params:
            p.grad = None
        with torch.no_grad():
            rec2, _ = ae_forward(sd, x, threshold)
        d_loss, _, _ = olo.discriminator_loss(sd_disc, x, rec2)
        d_loss.backward()
        opt_d.step()
        out.append((float(r["loss"].detach()), float(d_loss.detach())))
    return out


def train_steps(sd, batches, threshold, lr=1e-4, steps=1):
    """sd: {name: tensor}; trains every floating tensor except the EMA codebook/buffers in place."""
    params = [v.requires_grad_(True) for k, v in sd.items()
              if v.dtype == torch.float32 and v.dim() > 0 and not k.startswith("quantize.")]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.5, 0.9))
    losses = []
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ae_loss(sd, batches[s % len(batches)], threshold)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses
