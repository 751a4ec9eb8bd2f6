"""Host-side mirror of the DQ-VAE training losses.

Mirrors /root/reference/modules/losses/vqperceptual_multidisc.py:25-194 (hinge losses, adaptive weight,
VQLPIPSWithDiscriminator), modules/losses/vqperceptual.py:9-11 (DummyLoss) and
modules/dynamic_modules/budget.py:4-60.  The L1 term runs on dvq_l1_loss; scalar bookkeeping is host-side.

Every term runs on the HIP library: L1 (dvq_l1_loss), LPIPS (VGG16 on the conv kernels + dvq_maxpool2x2* /
dvq_lpips_head), the PatchGAN (4x4 convs on the implicit-GEMM kernels, BatchNorm+LeakyReLU on the normalisation
kernels) and the adaptive generator weight (two conv_out wgrad calls).  Only the reductions over the
[B,1,30,30] logit maps and the scalar bookkeeping are host-side torch.  LPIPS needs ImageNet VGG16 weights that
cannot be obtained offline: its parameters are random unless a checkpoint is loaded ("parity unpinned" for the
VALUE of the perceptual term with real weights; the computation itself is pinned with deterministic weights).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import kernels as K
from . import runtime as rt
from .config import instantiate_from_config
from .layers import ActNorm, BatchNorm2d, Conv2d, HipModule, Tape, _child


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


class NLayerDiscriminator(HipModule):
    """PatchGAN discriminator (modules/discriminator/model.py:17-67): same Sequential indices / parameter names as
    the reference (`main.0.weight`, `main.3.running_mean`, ...), compute on the HIP kernels:
      4x4 convs (stride 2,2,..,1,1; pad 1) on the implicit-GEMM kernels, the first LeakyReLU fused into its conv's
      epilogue, BatchNorm+LeakyReLU on the normalisation kernels (one group per channel over the whole batch)."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        norm_layer = ActNorm if use_actnorm else BatchNorm2d          # model.py:30-37: ActNorm layers follow convs WITH a bias
        use_bias = use_actnorm
        kw, padw = 4, 1
        seq = [Conv2d(input_nc, ndf, kw, 2, padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kw, 2, padw, bias=use_bias), norm_layer(ndf * nf_mult),
                    nn.LeakyReLU(0.2, True)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kw, 1, padw, bias=use_bias), norm_layer(ndf * nf_mult),
                nn.LeakyReLU(0.2, True)]
        seq += [Conv2d(ndf * nf_mult, 1, kw, 1, padw)]
        self.main = nn.Sequential(*seq)
        self.input_nc = input_nc
        # execution plan: (kind, module index, fused activation)
        plan, mods, i = [], list(self.main), 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, Conv2d) and isinstance(nxt, nn.LeakyReLU):
                plan.append(("conv", i, K.ACT_LRELU))
                i += 2
            elif isinstance(m, Conv2d):
                plan.append(("conv", i, K.ACT_NONE))
                i += 1
            elif isinstance(m, (BatchNorm2d, ActNorm)):
                assert isinstance(nxt, nn.LeakyReLU)
                plan.append(("bn", i, K.ACT_LRELU))
                i += 2
            else:
                raise AssertionError(type(m))
        self._plan = plan

    def fwd(self, x, tape):
        """x: NHWC [N,H,W,pad(input_nc)] -> logits NHWC [N,h,w,pad(1)] (channel 0 is the logit)"""
        h = x
        for kind, i, act in self._plan:
            m = self.main[i]
            h = m.fwd(h, _child(tape, str(i)), act=act)
        return h

    def bwd(self, dlogits, tape, need_dw=True, need_dx=True):
        g = dlogits
        for pi in range(len(self._plan) - 1, -1, -1):
            kind, i, act = self._plan[pi]
            m = self.main[i]
            t = tape.child(str(i))
            if kind == "bn":
                g = m.bwd(g, t, need_dw=need_dw)
                continue
            first = pi == 0
            prev = self._plan[pi - 1] if not first else None
            fused_prev = prev is not None and prev[0] == "conv" and prev[2] != K.ACT_NONE
            if first and not need_dx:
                m.bwd(g, t, need_dx=False, need_dw=need_dw)
                return None
            g = m.bwd(g, t, need_dw=need_dw, mask=t.s["x"] if fused_prev else None,
                      mask_act=prev[2] if fused_prev else K.ACT_NONE)
        return g

    # NCHW fp32 adapters (reference call signature: logits [B,1,h,w])
    def _fwd_nchw(self, x, tape):
        cd = rt.compute_dtype()
        y = self.fwd(K.nchw_to_nhwc_pad(x.contiguous().float(), _padc(self.input_nc, cd), cd), tape)
        return K.nhwc_pad_to_nchw(y, 1)

    def _bwd_nchw(self, dy, tape, in_dtype):
        cd = rt.compute_dtype()
        dx = self.bwd(K.nchw_to_nhwc_pad(dy.contiguous().float(), _padc(1, cd), cd), tape)
        return K.nhwc_pad_to_nchw(dx, self.input_nc)


def _padc(c, dtype):
    v = K.vec(dtype)
    return -(-c // v) * v


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


# ---- LPIPS ---------------------------------------------------------------------------------------------------
class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])


class NetLinLayer(nn.Module):
    """parameter holder for the 1x1 "lin" conv (lpips.py:64-70); the Dropout keeps the reference's Sequential index"""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class vgg16(nn.Module):
    """torchvision VGG16 feature stack cut into the five LPIPS slices (lpips.py:72-110), conv modules registered
    under torchvision's indices so `net.slice3.12.weight` etc. load.  torchvision / ImageNet weights are not
    available offline: parameters are random (He) unless a checkpoint is loaded."""

    SLICES = (("slice1", (0, 2), (3, 64, 64)), ("slice2", (5, 7), (64, 128, 128)), ("slice3", (10, 12, 14), (128, 256, 256, 256)),
              ("slice4", (17, 19, 21), (256, 512, 512, 512)), ("slice5", (24, 26, 28), (512, 512, 512, 512)))

    def __init__(self, requires_grad=False, pretrained=True):
        super().__init__()
        self.N_slices = 5
        for name, idxs, chans in self.SLICES:
            seq = nn.Sequential()
            for j, i in enumerate(idxs):
                conv = Conv2d(chans[j], chans[j + 1], 3, 1, 1)
                nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
                nn.init.zeros_(conv.bias)
                seq.add_module(str(i), conv)
            setattr(self, name, seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def convs(self):
        return [[getattr(self, name)[j] for j in range(len(idxs))] for name, idxs, _ in self.SLICES]


_GEN_SIDE = os.environ.get("DVQ_GEN_SIDE", "1") != "0"
_LOSS_PREFETCH = os.environ.get("DVQ_LOSS_PREFETCH", "1") != "0"


class LPIPS(nn.Module):
    """Learned perceptual metric (modules/losses/lpips.py:11-50) on the HIP kernels.

    The target and the reconstruction run through VGG16 as ONE batch of 2B images (conv + bias + ReLU in one
    kernel); every tap's normalise / difference / lin / spatial-mean is one kernel that also emits the gradient
    w.r.t. the reconstruction's features; the backward walks the reconstruction half only (dgrad kernels gated by the
    ReLU masks, max-pool routing fused with the tap gradient).  VGG16 and the lin layers are frozen, as in the reference.
    NetLinLayer dropout: the reference leaves nn.Dropout(0.5) on the squared differences active in training (Lightning's
    model.train() reaches the loss module; lpips.py:64-70).  `lin_dropout=True` (or DVQ_LPIPS_DROPOUT=1) applies it here -- in
    training mode only, hash-seeded per call inside the head kernel, so the training distribution matches the reference's; the
    draws are device-RNG dependent (parity unpinned), the default (off) computes the expectation of that value."""

    def __init__(self, use_dropout=True, lin_dropout=None):
        super().__init__()
        import os
        self.lin_dropout = bool(use_dropout) and (os.environ.get("DVQ_LPIPS_DROPOUT", "0") == "1" if lin_dropout is None else bool(lin_dropout))
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = vgg16(pretrained=True, requires_grad=False)
        for k, c in enumerate(self.chns):
            setattr(self, f"lin{k}", NetLinLayer(c, use_dropout=use_dropout))
            lin = getattr(self, f"lin{k}").model[-1]
            nn.init.uniform_(lin.weight, 0.0, 2.0 / c)
        for p in self.parameters():
            p.requires_grad = False
        self._aff = None
        self.pretrained_loaded = self._try_load_pretrained()

    # -- pretrained weights (lpips.py:16, 23-27 + torchvision vgg16(pretrained=True), lpips.py:79) -----------------------------
    ENV_VGG, ENV_LIN = "DVQ_VGG16_WEIGHTS", "DVQ_LPIPS_LIN_WEIGHTS"

    def _try_load_pretrained(self) -> bool:
        """The reference builds LPIPS from torchvision's ImageNet VGG16 and the `vgg.pth` lin layers; neither can be fetched
        offline, so they are taken from files: $DVQ_VGG16_WEIGHTS (torchvision `vgg16-397923af.pth`, keys `features.N.*`) and
        $DVQ_LPIPS_LIN_WEIGHTS (`vgg.pth`, keys `linK.model.1.weight`), else the usual cache locations.  Returns whether BOTH
        were loaded; without them the perceptual term is computed with random features (the caller warns)."""
        import os
        vgg_candidates = [os.environ.get(self.ENV_VGG), os.path.expanduser("~/.cache/torch/hub/checkpoints/vgg16-397923af.pth")]
        lin_candidates = [os.environ.get(self.ENV_LIN), "modules/lpips/vgg.pth",
                          os.path.expanduser("~/.cache/taming/modules/autoencoder/lpips/vgg.pth")]
        vgg = next((c for c in vgg_candidates if c and os.path.isfile(c)), None)
        lin = next((c for c in lin_candidates if c and os.path.isfile(c)), None)
        if vgg is None or lin is None:
            return False
        self.load_pretrained(vgg, lin)
        return True

    @torch.no_grad()
    def load_pretrained(self, vgg16_path, lin_path):
        """copy torchvision VGG16 feature weights and the LPIPS lin layers into this module (shapes checked, strict)"""
        sd = torch.load(vgg16_path, map_location="cpu", weights_only=True)
        for name, idxs, _ in vgg16.SLICES:
            seq = getattr(self.net, name)
            for i in idxs:
                conv = seq[[str(j) for j in idxs].index(str(i))]
                w, b = sd[f"features.{i}.weight"], sd[f"features.{i}.bias"]
                if tuple(w.shape) != tuple(conv.weight.shape):
                    raise ValueError(f"features.{i}.weight: {tuple(w.shape)} != {tuple(conv.weight.shape)}")
                conv.weight.copy_(w)
                conv.bias.copy_(b)
        lin_sd = torch.load(lin_path, map_location="cpu", weights_only=True)
        for k in range(len(self.chns)):
            w = lin_sd[f"lin{k}.model.1.weight"]
            tgt = getattr(self, f"lin{k}").model[-1].weight
            if tuple(w.shape) != tuple(tgt.shape):
                raise ValueError(f"lin{k}.model.1.weight: {tuple(w.shape)} != {tuple(tgt.shape)}")
            tgt.copy_(w)
        rt.bump_weights_epoch()
        self.pretrained_loaded = True

    def _affine(self, dtype, device):
        key = (dtype, device, self.scaling_layer.scale.data_ptr())
        if self._aff is None or self._aff[0] != key:
            cp = _padc(3, dtype)
            a = torch.zeros(cp, dtype=torch.float32, device=device)
            b = torch.zeros(cp, dtype=torch.float32, device=device)
            sc = self.scaling_layer.scale.reshape(3).to(device)
            sh = self.scaling_layer.shift.reshape(3).to(device)
            a[:3] = 1.0 / sc
            b[:3] = -sh / sc
            self._aff = (key, a, b)
        return self._aff[1], self._aff[2]

    def _trunk(self, h):
        """VGG16 slices on the scaled image batch h -> per slice the list of post-ReLU outputs"""
        acts = []
        for si, convs in enumerate(self.net.convs()):
            if si > 0:
                h = K.maxpool2x2(h)
            outs = []
            for conv in convs:
                h = conv.fwd(h, None, act=K.ACT_RELU)
                outs.append(h)
            acts.append(outs)
        return acts

    def target_features(self, x_p):
        """the five tap activations of the TARGET images alone (they do not depend on the autoencoder: VQLPIPSWithDiscriminator
        issues this on the side stream before the autoencoder's forward and hands the result to fwd(target_taps=...))"""
        a_c, b_c = self._affine(x_p.dtype, x_p.device)
        return [outs[-1] for outs in self._trunk(K.affine_channels(x_p, a_c, b_c))]

    def fwd(self, x_p, r_p, gscale=None, target_taps=None):
        """x_p / r_p: NHWC channel-padded images (target, reconstruction).  Returns (val fp32 [B], d_r) where
        d_r = gscale * d sum_b(val[b]) / d r_p when gscale is given, else None.
        target_taps: target_features(x_p) computed earlier -- the trunk then runs on the reconstruction alone (same kernels on B
        instead of 2 B images; every image's values are independent of the batch it is in)."""
        b = r_p.shape[0]
        a_c, b_c = self._affine(r_p.dtype, r_p.device)
        want = gscale is not None
        if target_taps is None:
            acts = self._trunk(K.affine_channels(torch.cat([x_p, r_p], 0), a_c, b_c))
            off = b               # rows of the reconstruction in the trunk's tensors
            taps_x = [outs[-1][:b] for outs in acts]
        else:
            acts = self._trunk(K.affine_channels(r_p, a_c, b_c))
            off = 0
            taps_x = target_taps
        val = K.zeros_small((b,), torch.float32, r_p.device)
        dtaps = []
        for k, outs in enumerate(acts):
            lin = getattr(self, f"lin{k}").model[-1].weight.reshape(-1)
            pd = getattr(self, f"lin{k}").model[0].p if (self.lin_dropout and self.training) else 0.0
            dtaps.append(K.lpips_head(taps_x[k], outs[-1][off:], lin, val, gscale if want else 0.0, want, p_drop=pd,
                                      seed=rt.next_dropout_seed() if pd > 0.0 else 0))
        if not want:
            return val, None
        g = None
        for si in range(4, -1, -1):
            convs, outs = self.net.convs()[si], acts[si]
            a_tap = outs[-1][off:]
            dz = dtaps[si] if si == 4 else K.maxpool2x2_relu_bwd(a_tap, dpool=g, dtap=dtaps[si])
            for j in range(len(convs) - 1, -1, -1):
                conv = convs[j]
                if j > 0:
                    src = outs[j - 1][off:]
                    d = conv._desc(src)
                    dz = K.conv2d_dgrad(d, dz, conv.packed(src.dtype)[1], mask=src, mask_act=K.ACT_RELU)
                else:
                    shp = list(outs[0].shape)
                    n, hh, ww = b, shp[1], shp[2]
                    cin_p = conv._padded(dz.dtype)[0]
                    d = K.conv_desc(n, hh, ww, cin_p, shp[3], 3, 3, 1, 1, 1, hh, ww, False, dz.dtype, rt.impl())
                    g = K.conv2d_dgrad(d, dz, conv.packed(dz.dtype)[1])
        return val, K.affine_channels(g, a_c, None)

    def forward(self, input, target):
        """reference signature: NCHW images -> [B,1,1,1] (no gradient: use VQLPIPSWithDiscriminator for training)"""
        cd = rt.compute_dtype()
        cp = _padc(3, cd)
        val, _ = self.fwd(K.nchw_to_nhwc_pad(input.contiguous().float(), cp, cd), K.nchw_to_nhwc_pad(target.contiguous().float(), cp, cd))
        return val.clone().view(-1, 1, 1, 1)


def vanilla_d_loss(logits_real, logits_fake):
    return 0.5 * (torch.mean(torch.nn.functional.softplus(-logits_real)) + torch.mean(torch.nn.functional.softplus(logits_fake)))


def _log(t, eps=1e-10):
    return torch.log(t + eps)


def bce_discr_loss(logits_real, logits_fake):
    return (-_log(1 - torch.sigmoid(logits_fake)) - _log(torch.sigmoid(logits_real))).mean()


def bce_gen_loss(logits_fake):
    return -_log(torch.sigmoid(logits_fake)).mean()


def _logit_loss_and_grad(fn, *logits_p):
    """Evaluate a reference GAN loss formula on the logit maps ([N,h,w,pad] NHWC tensors whose channel 0 is the logit;
    a few thousand elements -- host-side torch like the rest of the scalar bookkeeping) and return
    (loss fp32 scalar, [d loss / d logits in the same padded layout])."""
    leaves = [lp[..., 0].float().detach().requires_grad_(True) for lp in logits_p]
    with torch.enable_grad():
        loss = fn(*leaves)
        grads = torch.autograd.grad(loss, leaves)
    out = []
    for lp, g in zip(logits_p, grads):
        gp = torch.zeros_like(lp)
        gp[..., 0] = g.to(lp.dtype)
        out.append(gp)
    return loss.detach(), out, [l.detach() for l in leaves]


class VQLPIPSWithDiscriminator(nn.Module):
    """modules/losses/vqperceptual_multidisc.py:44-194 on the HIP path.

    optimizer_idx 0 (generator):  nll = mean(|x - xrec| + pw * LPIPS) ; g = -mean(D(xrec)) ;
        d_weight = clamp(|d nll / d W_last| / (|d g / d W_last| + 1e-4), 0, 1e4) * disc_weight (<= disc_weight_max) ;
        loss = nll + d_weight * disc_factor * g + codebook_weight * qloss [+ budget].
      The two last-layer gradients are one wgrad call each on the decoder's conv_out (the decoder publishes that
      closure as `last_layer._dvq_wgrad`); d_weight stays on the device, the combined gradient w.r.t. xrec is formed
      by one axpy kernel and handed to the autoencoder backward.  Discriminator / VGG parameters get no gradient.
    optimizer_idx 1 (discriminator): d_loss = disc_factor * hinge(D(x), D(xrec.detach())), two separate D passes
      (separate BatchNorm statistics) exactly like the reference.
    """

    def __init__(self, disc_start, disc_config, disc_init, codebook_weight=1.0, pixelloss_weight=1.0, disc_factor=1.0,
                 disc_weight=1.0, perceptual_weight=1.0, disc_conditional=False, disc_adaptive_loss=True,
                 disc_loss="hinge", disc_weight_max=None, budget_loss_config=None):
        super().__init__()
        assert disc_loss in ["hinge", "vanilla", "bce"]
        self.codebook_weight, self.pixel_weight = codebook_weight, pixelloss_weight
        self.perceptual_weight = perceptual_weight
        if perceptual_weight > 0:
            self.perceptual_loss = LPIPS().eval()
            if not self.perceptual_loss.pretrained_loaded:
                import warnings
                warnings.warn(
                    "VQLPIPSWithDiscriminator: perceptual_weight > 0 but no pretrained LPIPS weights were found -- the perceptual "
                    "term (and with it the adaptive discriminator weight) is computed with a RANDOM frozen VGG16.  Point "
                    f"${LPIPS.ENV_VGG} at torchvision's vgg16-397923af.pth and ${LPIPS.ENV_LIN} at the LPIPS vgg.pth (or load a "
                    "checkpoint that carries loss.perceptual_loss.*) before training for quality.", stacklevel=2)
        self.discriminator_iter_start = disc_start
        self.discriminator = instantiate_from_config(disc_config)
        if disc_init:
            self.discriminator = self.discriminator.apply(weights_init)
        self.disc_loss_name = disc_loss
        self.disc_loss = {"hinge": hinge_d_loss, "vanilla": vanilla_d_loss, "bce": bce_discr_loss}[disc_loss]
        self.gen_loss = bce_gen_loss if disc_loss == "bce" else hinge_g_loss
        self.disc_factor, self.discriminator_weight = disc_factor, disc_weight
        self.disc_conditional, self.disc_adaptive_loss, self.disc_weight_max = disc_conditional, disc_adaptive_loss, disc_weight_max
        if disc_conditional:
            # the reference asserts `not self.disc_conditional` whenever cond is None (vqperceptual_multidisc.py:121-129), and none of its
            # models passes a cond: the option cannot run there either
            raise NotImplementedError("disc_conditional=True needs a `cond` input that no model of the reference provides")
        self.budget_loss_config = budget_loss_config
        if budget_loss_config is not None:
            self.budget_loss = instantiate_from_config(budget_loss_config)

    # -- generator branch ---------------------------------------------------------------------------------------
    def prefetch_targets(self, inputs, optimizer_idx, global_step):
        """The part of this step's loss that depends on the TARGET images only -- generator step: their VGG16 tap activations;
        discriminator step: the PatchGAN forward of the real images -- issued on the side stream BEFORE the autoencoder's forward
        (models call this at the top of training_step), so that it runs beside the autoencoder's HBM-bound passes instead of after
        them.  forward() picks the results up if it is then called with the same tensor; DVQ_LOSS_PREFETCH=0 switches it off."""
        self._pre = None
        if not (_LOSS_PREFETCH and rt.side_wgrad_enabled() and torch.is_grad_enabled() and self.training and inputs.is_cuda):
            return
        x = inputs.contiguous().float()
        cd = rt.compute_dtype()
        cp = _padc(3, cd)
        st = {"key": (x.data_ptr(), tuple(x.shape), int(optimizer_idx), cd), "x": x}
        if optimizer_idx == 0 and self.perceptual_weight > 0:
            def work():
                st["x_p"] = K.nchw_to_nhwc_pad(x, cp, cd)
                st["taps"] = self.perceptual_loss.target_features(st["x_p"])
        elif optimizer_idx == 1 and any(p.requires_grad for p in self.discriminator.parameters()):
            def work():
                st["t_real"] = Tape()
                st["lr"] = self.discriminator.fwd(K.nchw_to_nhwc_pad(x, cp, cd), st["t_real"])
        else:
            return
        rt.run_on_side(work, x)
        self._pre = st

    def _take_prefetched(self, x, optimizer_idx):
        """the record prefetch_targets left for exactly this call (the current stream then waits for the side stream), else None"""
        st, self._pre = getattr(self, "_pre", None), None
        if st is None or st["key"] != (x.data_ptr(), tuple(x.shape), int(optimizer_idx), rt.compute_dtype()):
            return None
        rt.join_side(x.device)
        return st

    def _generator(self, x, xrec, want_grad, last_layer, disc_factor):
        """-> dict(nll, p, g, d_weight [device scalars], g_rec NCHW fp32 or None)"""
        cd = rt.compute_dtype()
        cp = _padc(3, cd)
        dev = x.device
        numel = x.numel()
        b = x.shape[0]
        pre = self._take_prefetched(x, 0) if want_grad else None
        l1_sum, g_l1 = K.l1_loss(x, xrec, scale_dev=torch.full((1,), 1.0 / numel, device=dev) if want_grad else None,
                                 want_grad=want_grad)
        rec_mean = (l1_sum / numel).to(torch.float32).reshape(())
        x_p = pre["x_p"] if pre is not None else K.nchw_to_nhwc_pad(x, cp, cd)
        r_p = K.nchw_to_nhwc_pad(xrec, cp, cd)
        out = {"rec": rec_mean}
        g_nll = K.nchw_to_nhwc_pad(g_l1, cp, cd) if want_grad else None
        # The GAN branch (PatchGAN forward + input-gradient backward of the reconstruction) and the perceptual branch (VGG16 forward +
        # backward) only share r_p: with both wanted, the GAN branch runs on the side stream so that its HBM-bound BatchNorm /
        # activation passes and the perceptual branch's pooling passes overlap the other branch's convolutions (DVQ_GEN_SIDE=0: off)
        gan_side = None
        if want_grad and disc_factor != 0 and self.perceptual_weight > 0 and _GEN_SIDE and rt.side_wgrad_enabled():
            gan_side = {}

            def gan_branch():
                tape_ = Tape()
                lf = self.discriminator.fwd(r_p, tape_)
                gl, (dl,), _ = _logit_loss_and_grad(self.gen_loss, lf)
                gan_side["g"] = gl
                gan_side["g_g"] = self.discriminator.bwd(dl, tape_, need_dw=False)

            rt.run_on_side(gan_branch, r_p)
        if self.perceptual_weight > 0:
            # nll = mean over B*3*H*W elements of (|x-xrec| + pw * p[b])  ->  d nll / d p[b] = pw / B
            val, d_r = self.perceptual_loss.fwd(x_p, r_p, gscale=self.perceptual_weight / b if want_grad else None,
                                                target_taps=pre["taps"] if pre is not None else None)
            out["p"] = val.mean()
            nll = rec_mean + self.perceptual_weight * out["p"]
            if want_grad:
                g_nll = K.add(g_nll, d_r)
        else:
            out["p"] = torch.zeros((), device=dev)
            nll = rec_mean
        out["nll"] = nll
        if disc_factor == 0:
            # the GAN term is multiplied by zero: skip the discriminator passes the reference would still run
            # (train_g_loss / train_d_weight are then logged as 0)
            out["g"] = torch.zeros((), device=dev)
            out["d_weight"] = torch.zeros((), device=dev)
            out["g_rec"] = K.nhwc_pad_to_nchw(g_nll, 3) if want_grad else None
            return out
        disc = self.discriminator
        if gan_side is not None:
            rt.join_side(dev)
            out["g"], g_g = gan_side["g"], gan_side["g_g"]
        else:
            tape = Tape() if want_grad else None
            logits_fake = disc.fwd(r_p, tape)
            g_loss, (dlog,), _ = _logit_loss_and_grad(self.gen_loss, logits_fake)
            out["g"] = g_loss
            if not want_grad:
                out["d_weight"] = torch.zeros((), device=dev)     # reference: autograd.grad fails in eval -> 0
                out["g_rec"] = None
                return out
            g_g = disc.bwd(dlog, tape, need_dw=False)
        if self.disc_adaptive_loss:
            hook = getattr(last_layer, "_dvq_wgrad", None)
            if hook is None:
                raise RuntimeError("adaptive discriminator weight needs the decoder's last-layer weight-gradient closure "
                                   "(`last_layer._dvq_wgrad`, published by DualGrainVQModel.ae_fwd)")
            pair = getattr(last_layer, "_dvq_wgrad_pair", None)
            nll_grads, g_grads = pair(g_nll, g_g) if pair is not None else (hook(g_nll), hook(g_g))
            n_nll, n_g = torch.linalg.vector_norm(nll_grads), torch.linalg.vector_norm(g_grads)
            self.last_adaptive_norms = (n_nll, n_g)          # device scalars, kept for diagnostics / precision tests
            d_weight = (n_nll / (n_g + 1e-4)).clamp_(0.0, 1e4)
            d_weight = d_weight * self.discriminator_weight
            if self.disc_weight_max is not None:
                d_weight = d_weight.clamp(max=self.disc_weight_max)
        else:
            d_weight = torch.full((), float(self.disc_weight_max), device=dev)
        out["d_weight"] = d_weight
        scale = (d_weight * disc_factor).to(torch.float32).reshape(1).contiguous()
        out["g_rec"] = K.nhwc_pad_to_nchw(K.axpy_dev(g_nll, g_g, scale), 3)
        return out

    def forward(self, codebook_loss, inputs, reconstructions, optimizer_idx, global_step, last_layer=None, cond=None,
                split="train", gate=None):
        assert cond is None, "disc_conditional is not supported"
        disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
        x = inputs.contiguous().float()
        if optimizer_idx == 0:
            want_grad = torch.is_grad_enabled() and reconstructions.requires_grad
            loss_main, parts = _GenLossFn.apply(self, want_grad, x, reconstructions, last_layer, disc_factor)
            loss = loss_main + self.codebook_weight * codebook_loss.mean()
            log = {}
            if gate is not None and self.budget_loss_config is not None:
                budget_loss = self.budget_loss(gate=gate)
                loss = loss + budget_loss
                log["{}_budget_loss".format(split)] = budget_loss.detach().mean()
            log.update({"{}_total_loss".format(split): loss.detach().mean(),
                        "{}_quant_loss".format(split): codebook_loss.detach().mean(),
                        "{}_nll_loss".format(split): parts["nll"],
                        "{}_rec_loss".format(split): parts["nll"],
                        "{}_p_loss".format(split): parts["p"],
                        "{}_d_weight".format(split): parts["d_weight"],
                        "{}_disc_factor".format(split): torch.tensor(disc_factor),
                        "{}_g_loss".format(split): parts["g"]})
            return loss, log
        if optimizer_idx == 1:
            params = [p for p in self.discriminator.parameters() if p.requires_grad]
            want_grad = torch.is_grad_enabled() and len(params) > 0
            d_loss, m_real, m_fake = _DiscLossFn.apply(self, want_grad, x, reconstructions.detach().contiguous().float(),
                                                       disc_factor, *params)
            log = {"{}_disc_loss".format(split): d_loss.detach().mean(),
                   "{}_logits_real".format(split): m_real,
                   "{}_logits_fake".format(split): m_fake}
            return d_loss, log


class _GenLossFn(torch.autograd.Function):
    """nll + d_weight * disc_factor * g_loss as one autograd node; the gradient w.r.t. the reconstruction is produced in
    the forward (the adaptive weight needs both backward passes anyway)."""

    @staticmethod
    def forward(ctx, mod, want_grad, x, xrec, last_layer, disc_factor):
        with torch.no_grad():
            out = mod._generator(x, xrec.contiguous().float(), want_grad, last_layer, disc_factor)
        ctx.g_rec = out.pop("g_rec")
        loss = out["nll"] + out["d_weight"] * disc_factor * out["g"]
        parts = {k: v.detach() for k, v in out.items()}     # plain python object: passed through, not tracked
        ctx.set_materialize_grads(False)
        return loss.reshape(()), parts

    @staticmethod
    def backward(ctx, g, _parts=None):
        if g is None or ctx.g_rec is None:
            return None, None, None, None, None, None
        return None, None, None, ctx.g_rec * g, None, None


class _DiscLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, want_grad, x, xrec, disc_factor, *params):
        cd = rt.compute_dtype()
        cp = _padc(3, cd)
        disc = mod.discriminator
        with torch.no_grad():
            pre = mod._take_prefetched(x, 1) if want_grad else None
            t_real, t_fake = (Tape(), Tape()) if want_grad else (None, None)
            if pre is not None:       # the real images' forward ran on the side stream beside the autoencoder (prefetch_targets)
                lr, t_real = pre["lr"], pre["t_real"]
            else:
                lr = disc.fwd(K.nchw_to_nhwc_pad(x, cp, cd), t_real)
            lf = disc.fwd(K.nchw_to_nhwc_pad(xrec, cp, cd), t_fake)
            loss, grads, leaves = _logit_loss_and_grad(lambda a, b: disc_factor * mod.disc_loss(a, b), lr, lf)
        ctx.state = (disc, t_real, t_fake, grads) if want_grad else None
        ctx.n = len(params)
        m_real, m_fake = leaves[0].mean(), leaves[1].mean()
        ctx.mark_non_differentiable(m_real, m_fake)
        return loss.reshape(()), m_real, m_fake

    @staticmethod
    def backward(ctx, g, *_unused):
        if ctx.state is not None and g is not None:
            disc, t_real, t_fake, grads = ctx.state
            with torch.no_grad():
                # the upstream scale (1 in the trainer) multiplies the tiny logit gradients on the device: no host sync
                for tape, dl in ((t_real, grads[0]), (t_fake, grads[1])):
                    disc.bwd((dl.float() * g).to(dl.dtype), tape, need_dw=True, need_dx=False)
        return (None,) * (5 + ctx.n)


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
