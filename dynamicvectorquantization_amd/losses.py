"""Host-side mirror of the DQ-VAE training losses.

Mirrors /root/reference/modules/losses/vqperceptual_multidisc.py:25-194 (hinge losses, adaptive weight,
VQLPIPSWithDiscriminator), modules/losses/vqperceptual.py:9-11 (DummyLoss) and
modules/dynamic_modules/budget.py:4-60.  The L1 term runs on dvq_l1_loss; scalar bookkeeping is host-side.

Round-1 status: the reconstruction (L1) + codebook terms are on the HIP path.  The PatchGAN discriminator
and LPIPS branches need two more kernel families (BatchNorm/LeakyReLU, VGG max-pool) and raise
NotImplementedError when their weights are non-zero; LPIPS additionally needs ImageNet VGG16 weights that
cannot be obtained offline ("parity unpinned").
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from .config import instantiate_from_config


class DummyLoss(nn.Module):
    def __init__(self):
        super().__init__()


def adopt_weight(weight, global_step, threshold=0, value=0.):
    if global_step < threshold:
        weight = value
    return weight


def hinge_d_loss(logits_real, logits_fake):
    loss_real = torch.mean(torch.relu(1. - logits_real))
    loss_fake = torch.mean(torch.relu(1. + logits_fake))
    return 0.5 * (loss_real + loss_fake)


def hinge_g_loss(logits_fake):
    return -torch.mean(logits_fake)


class _L1MeanFn(torch.autograd.Function):
    """mean |x - xrec| over all elements (vqperceptual_multidisc.py:116,126); gradient w.r.t. xrec only."""

    @staticmethod
    def forward(ctx, x, xrec):
        x = x.contiguous()
        xrec = xrec.contiguous()
        ctx.save_for_backward(x, xrec)
        loss_sum, _ = K.l1_loss(x, xrec)
        return (loss_sum / x.numel()).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        x, xrec = ctx.saved_tensors
        scale = (g.to(torch.float32) / x.numel()).reshape(1).contiguous()
        _, grad = K.l1_loss(x, xrec, scale_dev=scale, want_grad=True)
        return None, grad


def l1_mean(x, xrec):
    return _L1MeanFn.apply(x, xrec)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator (modules/discriminator/model.py:17-67): parameters with the reference's names so
    checkpoints load; its compute (4x4 convs + BatchNorm + LeakyReLU) is not on the HIP path yet."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("use_actnorm=True is unused by the shipped configs")
        kw, padw = 4, 1
        seq = [nn.Conv2d(input_nc, ndf, kw, 2, padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kw, 2, padw, bias=False), nn.BatchNorm2d(ndf * nf_mult),
                    nn.LeakyReLU(0.2, True)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kw, 1, padw, bias=False), nn.BatchNorm2d(ndf * nf_mult),
                nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * nf_mult, 1, kw, 1, padw)]
        self.main = nn.Sequential(*seq)

    def forward(self, input):
        raise NotImplementedError("PatchGAN forward has no HIP path yet (round-2 scope, SURVEY K17)")


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class VQLPIPSWithDiscriminator(nn.Module):
    def __init__(self, disc_start, disc_config, disc_init, codebook_weight=1.0, pixelloss_weight=1.0, disc_factor=1.0,
                 disc_weight=1.0, perceptual_weight=1.0, disc_conditional=False, disc_adaptive_loss=True,
                 disc_loss="hinge", disc_weight_max=None, budget_loss_config=None):
        super().__init__()
        assert disc_loss in ["hinge", "vanilla", "bce"]
        self.codebook_weight, self.pixel_weight = codebook_weight, pixelloss_weight
        self.perceptual_weight = perceptual_weight
        self.discriminator_iter_start = disc_start
        self.discriminator = instantiate_from_config(disc_config)
        if disc_init:
            self.discriminator = self.discriminator.apply(weights_init)
        self.disc_loss_name = disc_loss
        self.disc_factor, self.discriminator_weight = disc_factor, disc_weight
        self.disc_conditional, self.disc_adaptive_loss, self.disc_weight_max = disc_conditional, disc_adaptive_loss, disc_weight_max
        self.budget_loss_config = budget_loss_config
        if budget_loss_config is not None:
            self.budget_loss = instantiate_from_config(budget_loss_config)

    def forward(self, codebook_loss, inputs, reconstructions, optimizer_idx, global_step, last_layer=None, cond=None,
                split="train", gate=None):
        rec_mean = l1_mean(inputs, reconstructions)
        if self.perceptual_weight > 0:
            raise NotImplementedError("LPIPS (perceptual_weight > 0) has no HIP path yet and its VGG16 weights are "
                                      "not obtainable offline; set model.params.lossconfig.params.perceptual_weight=0")
        p_loss = torch.zeros((), device=inputs.device)
        nll_loss = rec_mean
        disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
        if disc_factor != 0:
            raise NotImplementedError("the PatchGAN branch (disc_factor != 0) has no HIP path yet; set "
                                      "model.params.lossconfig.params.disc_factor=0")
        if optimizer_idx == 0:
            d_weight = torch.zeros((), device=inputs.device)
            g_loss = torch.zeros((), device=inputs.device)
            loss = nll_loss + self.codebook_weight * codebook_loss.mean()
            log = {}
            if gate is not None and self.budget_loss_config is not None:
                budget_loss = self.budget_loss(gate=gate)
                loss = loss + budget_loss
                log["{}_budget_loss".format(split)] = budget_loss.detach().mean()
            log.update({"{}_total_loss".format(split): loss.clone().detach().mean(),
                        "{}_quant_loss".format(split): codebook_loss.detach().mean(),
                        "{}_nll_loss".format(split): nll_loss.detach().mean(),
                        "{}_rec_loss".format(split): rec_mean.detach(),
                        "{}_p_loss".format(split): p_loss.detach().mean(),
                        "{}_d_weight".format(split): d_weight.detach(),
                        "{}_disc_factor".format(split): torch.tensor(disc_factor),
                        "{}_g_loss".format(split): g_loss.detach().mean()})
            return loss, log
        if optimizer_idx == 1:
            d_loss = torch.zeros((), device=inputs.device, requires_grad=True)
            log = {"{}_disc_loss".format(split): d_loss.clone().detach().mean()}
            return d_loss, log


class BudgetConstraint_RatioMSE_DualGrain(nn.Module):
    """budget.py:4-28 (incl. calculate_all returning loss_last + loss_last)."""

    def __init__(self, target_ratio=0., gamma=1.0, min_grain_size=8, max_grain_size=16, calculate_all=True):
        super().__init__()
        self.target_ratio, self.gamma, self.calculate_all = target_ratio, gamma, calculate_all
        self.const = min_grain_size * min_grain_size
        self.max_const = max_grain_size * max_grain_size - self.const

    def forward(self, gate):
        gate = gate.float()
        beta = 1.0 * gate[:, 0, :, :] + 4.0 * gate[:, 1, :, :]
        beta = (beta.sum() / gate.size(0)) - self.const
        budget_ratio = beta / self.max_const
        loss_budget = self.gamma * (budget_ratio - self.target_ratio) ** 2
        if self.calculate_all:
            loss_budget_last = self.gamma * ((1 - budget_ratio) - (1 - self.target_ratio)) ** 2
            return loss_budget_last + loss_budget_last
        return loss_budget


class BudgetConstraint_NormedSeperateRatioMSE_TripleGrain(nn.Module):
    """budget.py:30-60."""

    def __init__(self, target_fine_ratio=0., target_median_ratio=0., gamma=1.0, min_grain_size=8, median_grain_size=16,
                 max_grain_size=32):
        super().__init__()
        assert target_fine_ratio + target_median_ratio <= 1.0
        self.target_fine_ratio, self.target_median_ratio, self.gamma = target_fine_ratio, target_median_ratio, gamma
        self.min_const = min_grain_size * min_grain_size
        self.median_const = median_grain_size * median_grain_size - self.min_const
        self.max_const = max_grain_size * max_grain_size - self.min_const

    def forward(self, gate):
        gate = gate.float()
        beta_median = 1.0 * gate[:, 0] + 4.0 * gate[:, 1] + 1.0 * gate[:, 2]
        ratio_median = ((beta_median.sum() / gate.size(0)) - self.min_const) / self.median_const
        loss_median = (ratio_median - self.target_median_ratio) ** 2
        beta_fine = 1.0 * gate[:, 0] + 16.0 * gate[:, 2] + 1.0 * gate[:, 1]
        ratio_fine = ((beta_fine.sum() / gate.size(0)) - self.min_const) / self.max_const
        loss_fine = self.gamma * (ratio_fine - self.target_fine_ratio) ** 2
        return loss_fine + loss_median
