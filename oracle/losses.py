"""Oracle: the training-loss networks of the DQ-VAE step (torch-CPU fp32, autograd for the gradients).
TEST INFRASTRUCTURE -- never imported by the product package.

Follows (behaviour, not code), all under /root/reference:
  * NLayerDiscriminator (PatchGAN)          modules/discriminator/model.py:17-67
  * LPIPS, ScalingLayer, NetLinLayer, vgg16 modules/losses/lpips.py:11-122   (torchvision VGG16 "D" feature stack:
                                            3x3 convs 64,64,M,128,128,M,256x3,M,512x3,M,512x3 with ReLU, taps after
                                            relu1_2 / 2_2 / 3_3 / 4_3 / 5_3)
  * hinge losses, adaptive weight, VQLPIPSWithDiscriminator.forward
                                            modules/losses/vqperceptual_multidisc.py:25-32,97-107,109-194

State dicts use the reference's key names ("main.0.weight", "net.slice1.0.weight", "lin0.model.1.weight", ...).
LPIPS is evaluated WITHOUT the NetLinLayer dropout (the reference builds `LPIPS().eval()` but Lightning's
`model.train()` re-enables that dropout during training; its mask is device-RNG dependent and has expectation equal to
the no-dropout value -- DESIGN.md "parity unpinned").
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

VGG_SLICES = (("slice1", (0, 2)), ("slice2", (5, 7)), ("slice3", (10, 12, 14)), ("slice4", (17, 19, 21)),
              ("slice5", (24, 26, 28)))
LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)


# ---- PatchGAN -------------------------------------------------------------------------------------------
def patchgan(sd, x, n_layers=3, train=True, prefix="main.", running=None):
    """returns logits [B,1,h,w]; BatchNorm in training mode uses batch statistics (biased variance, eps 1e-5);
    `running`: optional dict receiving the updated running_mean / running_var (momentum 0.1, unbiased variance)"""
    h = F.leaky_relu(F.conv2d(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.conv2d(h, sd[f"{prefix}{idx}.weight"], sd.get(f"{prefix}{idx}.bias"), stride=stride, padding=1)
        bn = f"{prefix}{idx + 1}"
        if bn + ".loc" in sd:                           # use_actnorm=True (utils/utils.py:58-110): h = scale * (x + loc), already initialised
            h = F.leaky_relu(sd[bn + ".scale"] * (h + sd[bn + ".loc"]), 0.2)
            idx += 3
            continue
        if train:
            mean = h.mean(dim=(0, 2, 3))
            var = h.var(dim=(0, 2, 3), unbiased=False)
            if running is not None:
                cnt = h.numel() / h.shape[1]
                rm = sd.get(bn + ".running_mean", torch.zeros_like(mean))
                rv = sd.get(bn + ".running_var", torch.ones_like(var))
                running[bn + ".running_mean"] = (0.9 * rm + 0.1 * mean).detach()
                running[bn + ".running_var"] = (0.9 * rv + 0.1 * var * cnt / (cnt - 1)).detach()
        else:
            mean, var = sd[bn + ".running_mean"], sd[bn + ".running_var"]
        h = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
        h = h * sd[bn + ".weight"][None, :, None, None] + sd[bn + ".bias"][None, :, None, None]
        h = F.leaky_relu(h, 0.2)
        idx += 3
    return F.conv2d(h, sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"], stride=1, padding=1)


# ---- LPIPS ----------------------------------------------------------------------------------------------
def vgg16_taps(sd, x, prefix="net."):
    taps = []
    h = x
    for si, (name, idxs) in enumerate(VGG_SLICES):
        if si > 0:
            h = F.max_pool2d(h, 2, 2)
        for i in idxs:
            h = F.relu(F.conv2d(h, sd[f"{prefix}{name}.{i}.weight"], sd[f"{prefix}{name}.{i}.bias"], padding=1))
        taps.append(h)
    return taps


def lpips(sd, x, xrec, prefix=""):
    """-> [B,1,1,1]"""
    shift = torch.tensor(LPIPS_SHIFT, dtype=x.dtype)[None, :, None, None]
    scale = torch.tensor(LPIPS_SCALE, dtype=x.dtype)[None, :, None, None]
    t0 = vgg16_taps(sd, (x - shift) / scale, prefix + "net.")
    t1 = vgg16_taps(sd, (xrec - shift) / scale, prefix + "net.")
    val = 0
    for k, (a, b) in enumerate(zip(t0, t1)):
        na = a / (torch.sqrt(torch.sum(a ** 2, dim=1, keepdim=True)) + 1e-10)
        nb = b / (torch.sqrt(torch.sum(b ** 2, dim=1, keepdim=True)) + 1e-10)
        d = (na - nb) ** 2
        val = val + F.conv2d(d, sd[f"{prefix}lin{k}.model.1.weight"]).mean(dim=(2, 3), keepdim=True)
    return val


# ---- loss assembly --------------------------------------------------------------------------------------
def hinge_d(lr, lf):
    return 0.5 * (torch.mean(F.relu(1. - lr)) + torch.mean(F.relu(1. + lf)))


def generator_loss(sd_disc, sd_lpips, x, xrec, qloss, last_layer, perceptual_weight=1.0, disc_factor=1.0,
                   disc_weight=1.0, disc_weight_max=None, codebook_weight=1.0, n_layers=3, running=None):
    """optimizer_idx == 0 branch (vqperceptual_multidisc.py:109-153).  xrec must be a function of `last_layer`
    (a leaf requiring grad).  Returns dict(loss, nll, p, g, d_weight, rec_mean).  `running`: optional dict receiving the
    BatchNorm running statistics this training-mode discriminator pass leaves (oracle.losses.patchgan)."""
    rec = torch.abs(x - xrec)
    if perceptual_weight > 0:
        p = lpips(sd_lpips, x, xrec)
        rec = rec + perceptual_weight * p
    else:
        p = torch.zeros(1)
    nll = rec.mean()
    logits_fake = patchgan(sd_disc, xrec, n_layers, running=running)
    g = -logits_fake.mean()
    ng = torch.autograd.grad(nll, last_layer, retain_graph=True)[0]
    gg = torch.autograd.grad(g, last_layer, retain_graph=True)[0]
    dw = (torch.norm(ng) / (torch.norm(gg) + 1e-4)).clamp(0.0, 1e4).detach() * disc_weight
    if disc_weight_max is not None:
        dw = dw.clamp(max=disc_weight_max)
    loss = nll + dw * disc_factor * g + codebook_weight * qloss.mean()
    return {"loss": loss, "nll": nll, "p": p, "g": g, "d_weight": dw, "rec_mean": rec.detach().mean()}


def discriminator_loss(sd_disc, x, xrec, disc_factor=1.0, n_layers=3, running=None):
    """optimizer_idx == 1 branch (vqperceptual_multidisc.py:170-188), hinge"""
    lr = patchgan(sd_disc, x.detach(), n_layers, running=running)
    sd2 = dict(sd_disc)
    if running is not None:
        sd2.update(running)         # the second call starts from the running statistics the first one left
    lf = patchgan(sd2, xrec.detach(), n_layers, running=running)
    return disc_factor * hinge_d(lr, lf), lr, lf
