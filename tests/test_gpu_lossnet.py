"""GPU parity of the training-loss networks (PatchGAN, LPIPS, VQLPIPSWithDiscriminator both optimizer branches) against
goldens captured from the reference (tests/golden/losses.npz, lossnet.npz) and against the oracle.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dynamicvectorquantization_amd import synth

pytestmark = pytest.mark.gpu


def _rel(got, ref, l2=False):
    """max-norm error relative to the largest reference entry; l2=True: relative Frobenius error (bf16 runs: a
    LeakyReLU / ReLU / max-pool decision that flips on a rounding-level pre-activation changes isolated gradient
    entries by O(1) of their size, which the max norm would report although the tensor as a whole agrees)"""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    if l2:
        return float(np.linalg.norm(got - ref)) / max(1e-30, float(np.linalg.norm(ref)))
    return float(np.abs(got - ref).max()) / max(1e-12, float(np.abs(ref).max()))


def _load_det(module, prefix, fn=synth.det_param):
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(torch.from_numpy(fn(prefix + n, tuple(p.shape))).to(p.device))
    from dynamicvectorquantization_amd import runtime as rt
    rt.bump_weights_epoch()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 6e-2)])
def test_patchgan_golden(dev, dtype, tol):
    """forward logits, input gradient, parameter gradients and BatchNorm running statistics vs the reference module"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.losses import NLayerDiscriminator
    g = load_golden("losses")
    with rt.compute_dtype_ctx(dtype):
        disc = NLayerDiscriminator(input_nc=3, ndf=16, n_layers=3).to(dev).train()
        _load_det(disc, "disc.")
        x = torch.from_numpy(synth.det_param("disc.x", (2, 3, 64, 64)) * 40).to(dev).requires_grad_(True)
        y = disc(x)
        assert tuple(y.shape) == g["disc_y"].shape
        gout = torch.from_numpy(synth.det_param("disc.gout", tuple(y.shape))).to(dev)
        (y * gout).sum().backward()
        l2 = dtype == torch.bfloat16
        assert _rel(y.detach().cpu().numpy(), g["disc_y"]) < tol
        assert _rel(x.grad.cpu().numpy(), g["disc_dx"], l2) < tol * 2
        for n, p in disc.named_parameters():
            assert _rel(p.grad.cpu().numpy(), g["disc_d." + n], l2) < tol * 2, n
        for n, b in disc.named_buffers():
            ref = g["disc_buf." + n]
            if ref.ndim == 0:
                assert int(b) == int(ref)
            else:
                np.testing.assert_allclose(b.cpu().numpy(), ref, rtol=max(tol, 1e-4), atol=tol * 1e-2)


# bf16 note: with the deterministic random VGG16 the normalised features of two images differ by only ~1e-2..1e-1 of
# their norm, so rounding the features to bf16 (2^-9 relative) perturbs the DIFFERENCE the metric is built on by tens
# of percent: bf16 rows are sanity bounds, the fp32 rows are the parity check.
@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-3, 2e-2), (torch.bfloat16, 0.6, 0.8)])
def test_lpips_golden(dev, dtype, tol, gtol):
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.losses import LPIPS, _padc
    g = load_golden("lossnet")
    with rt.compute_dtype_ctx(dtype):
        lp = LPIPS().to(dev)
        own = {k: tuple(v.shape) for k, v in lp.state_dict().items()}
        ref = {str(k): tuple(int(x) for x in str(s).split(",")) for k, s in zip(g["lpips_shapes_keys"], g["lpips_shapes"])}
        assert own == ref                                       # state_dict layout of the reference's LPIPS
        _load_det(lp, "lpips.", synth.det_lpips_param)
        cp = _padc(3, dtype)
        x_p = K.nchw_to_nhwc_pad(torch.from_numpy(g["lpips_x"]).to(dev), cp, dtype)
        r_p = K.nchw_to_nhwc_pad(torch.from_numpy(g["lpips_xrec"]).to(dev), cp, dtype)
        val, d_r = lp.fwd(x_p, r_p, gscale=1.0)
        assert _rel(val.cpu().numpy(), g["lpips_val"].reshape(-1)) < tol
        d = K.nhwc_pad_to_nchw(d_r, 3).cpu().numpy()
        assert _rel(d, g["lpips_dxrec"], True) < gtol
        v2 = lp(torch.from_numpy(g["lpips_x"]).to(dev), torch.from_numpy(g["lpips_xrec"]).to(dev))
        assert tuple(v2.shape) == (2, 1, 1, 1) and _rel(v2.cpu().numpy(), g["lpips_val"]) < tol


def test_lpips_lin_dropout_kernel(dev):
    """NetLinLayer dropout inside the head kernel (dvq_lpips_head_drop; the reference: lpips.py:64-70, active in its training
    mode).  The draws are device-RNG dependent (parity unpinned), so the checks are properties: p = 0 is the plain head bit for bit;
    a seed reproduces its value; the mean over seeds is the undropped value (inverted dropout); the gradient the kernel returns is
    the derivative of the SAME masked value (directional finite difference with the seed held)."""
    from dynamicvectorquantization_amd import kernels as K
    torch.manual_seed(11)
    n, hw, c = 2, 48 * 48, 128
    f0 = torch.relu(torch.randn(n, 48, 48, c, device=dev))
    f1 = torch.relu(torch.randn(n, 48, 48, c, device=dev))
    lin = torch.rand(c, device=dev) * (2.0 / c)

    def head(f1_, p, seed, want=False):
        val = torch.zeros(n, device=dev)
        d = K.lpips_head(f0, f1_, lin, val, 1.0 if want else 0.0, want, p_drop=p, seed=seed)
        return val, d

    v0, d0 = head(f1, 0.0, 0, True)
    vp, dp = head(f1, 0.0, 123, True)
    close = lambda a, b: float(((a - b).abs() / b.abs()).max()) < 1e-5       # (val is folded with atomics: equal up to summation order)
    assert close(v0, vp) and torch.equal(d0, dp)
    va, da = head(f1, 0.5, 7, True)
    vb, _ = head(f1, 0.5, 7)
    vc, _ = head(f1, 0.5, 8)
    assert close(va, vb) and not close(va, vc)
    mean = torch.stack([head(f1, 0.5, 1000 + s)[0] for s in range(64)]).mean(0)
    assert float(((mean - v0).abs() / v0).max()) < 0.01, (mean, v0)
    # (a dropped element still receives gradient through the channel normalisation of its pixel: no zero pattern to look for --
    #  the finite difference below is the check)
    # along the gradient itself (a random direction has a directional derivative of ~1e-7: each pixel's term is scale invariant and the
    # value is folded with fp32 atomics, so the difference quotient would be noise): u = da / |da|, derivative = |da|
    u = (da.double() / da.double().norm()).float()
    eps = 0.5            # (|da| ~ 7e-5: the values move by ~3.5e-5 each way, three orders above their summation noise; u's entries are < 0.02)
    fd = (head(f1 + eps * u, 0.5, 7)[0].double().sum() - head(f1 - eps * u, 0.5, 7)[0].double().sum()) / (2 * eps)
    an = (da.double() * u.double()).sum()
    assert float(an) > 2e-5 and abs(float(fd - an)) < 3e-2 * abs(float(an)), (float(fd), float(an))


def test_lpips_module_lin_dropout_switch(dev):
    """LPIPS(lin_dropout=True) drops only in training mode; the default module never does"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.losses import LPIPS, _padc
    g = load_golden("lossnet")
    with rt.compute_dtype_ctx(torch.float32):
        lp = LPIPS(lin_dropout=True).to(dev)
        _load_det(lp, "lpips.", synth.det_lpips_param)
        cp = _padc(3, torch.float32)
        x_p = K.nchw_to_nhwc_pad(torch.from_numpy(g["lpips_x"]).to(dev), cp, torch.float32)
        r_p = K.nchw_to_nhwc_pad(torch.from_numpy(g["lpips_xrec"]).to(dev), cp, torch.float32)
        lp.eval()
        v_eval = lp.fwd(x_p, r_p)[0].clone()
        assert _rel(v_eval.cpu().numpy(), g["lpips_val"].reshape(-1)) < 2e-3
        lp.train()
        v_a, v_b = lp.fwd(x_p, r_p)[0].clone(), lp.fwd(x_p, r_p)[0].clone()
        far = lambda a, b: float(((a - b).abs() / b.abs()).max()) > 1e-4
        assert far(v_a, v_b) and far(v_a, v_eval)                                     # a fresh mask per call
        assert float(((v_a - v_eval).abs() / v_eval).max()) < 0.25


def test_maxpool_and_head_kernels_vs_torch(dev):
    """kernel-level checks against a plain fp32 torch restatement (tie routing, ReLU gate, tap gradient)"""
    from dynamicvectorquantization_amd import kernels as K
    torch.manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16):
        a = torch.relu(torch.randn(2, 8, 12, 64, device=dev)).to(dtype)
        a[0, :2, :2, :8] = 0.5                 # a window of exact ties: the first position must receive the gradient
        y = K.maxpool2x2(a)
        ref = torch.nn.functional.max_pool2d(a.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        assert torch.equal(y.float(), ref)
        dpool = torch.randn_like(y.float()).to(dtype)
        dtap = torch.randn_like(a.float()).to(dtype)
        dz = K.maxpool2x2_relu_bwd(a, dpool, dtap)
        af = a.float().permute(0, 3, 1, 2).requires_grad_(True)
        torch.nn.functional.max_pool2d(af, 2, 2).backward(dpool.float().permute(0, 3, 1, 2))
        want = (af.grad.permute(0, 2, 3, 1) + dtap.float()) * (a.float() > 0)
        tol = 0 if dtype == torch.float32 else 2e-2
        assert float((dz.float() - want).abs().max()) <= tol * float(want.abs().max()) + 1e-12
        assert float(dz.float()[0, 0, 0, :8].abs().sum()) > 0 and float((dz.float()[0, 0, 1, :8] - dtap.float()[0, 0, 1, :8]).abs().max()) <= 1e-2


def _toy_last_layer(dev, dtype, g):
    """feat -> conv3x3(8 -> 3) = reconstruction, with the weight-gradient closure the loss module expects"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape, to_nhwc
    conv = Conv2d(8, 3, 3, 1, 1).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.det_param("lossnet.last.weight", (3, 8, 3, 3)) * 2.0))
        conv.bias.copy_(torch.from_numpy(synth.det_param("lossnet.last.bias", (3,))))
    rt.bump_weights_epoch()
    feat = torch.from_numpy(synth.det_param("lossnet.feat", (2, 8, 64, 64)) * 4.0).to(dev)
    tape = Tape()
    rec_p = conv.fwd(to_nhwc(feat, dtype), tape)
    xrec = K.nhwc_pad_to_nchw(rec_p, 3).requires_grad_(True)

    def wgrad(g_p):
        buf = torch.zeros(3, 8, 3, 3, device=dev)
        K.conv2d_wgrad_oihw(tape.s["d"], tape.s["x"], g_p, 8, 3, buf, None)
        return buf

    conv.weight._dvq_wgrad = wgrad
    return conv, tape, xrec


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 3e-3, 1e-2), (torch.bfloat16, 3e-2, 0.35)])
def test_vqlpips_with_discriminator_golden(dev, dtype, tol, gtol):
    """both optimizer branches of the reference loss on a toy last layer: loss values, adaptive weight (free and
    clamped), gradients w.r.t. the features / last layer, discriminator gradients and BatchNorm buffers"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden("lossnet")
    x = torch.from_numpy(synth.half_flat_images(2, 64, 16, seed=5)).to(dev)
    qloss = torch.tensor(0.123, device=dev)
    with rt.compute_dtype_ctx(dtype):
        for tag, wmax in (("free", None), ("capped", 0.005)):
            loss_mod = instantiate_from_config({"target": "modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator", "params": dict(
                disc_start=0, disc_init=True, disc_conditional=False, disc_loss="hinge", disc_factor=1.0, disc_weight=1.0,
                disc_weight_max=wmax, codebook_weight=1.0, pixelloss_weight=1.0, perceptual_weight=1.0,
                disc_config={"target": "modules.discriminator.model.NLayerDiscriminator",
                             "params": dict(input_nc=3, ndf=16, n_layers=3, use_actnorm=False)})}).to(dev).train()
            _load_det(loss_mod.discriminator, "disc.")
            _load_det(loss_mod.perceptual_loss, "lpips.", synth.det_lpips_param)
            conv, tape, xrec = _toy_last_layer(dev, dtype, g)
            if dtype == torch.float32:
                assert _rel(xrec.detach().cpu().numpy(), g["gen_xrec"]) < 1e-3
            loss, log = loss_mod(qloss, x, xrec, 0, 0, last_layer=conv.weight, split="train")
            loss.backward()
            for k in ("nll_loss", "p_loss", "g_loss"):
                np.testing.assert_allclose(float(log["train_" + k]), float(g[f"gen_{tag}_{k}"]), rtol=tol, atol=tol * 1e-2)
            np.testing.assert_allclose(float(log["train_d_weight"]), float(g[f"gen_{tag}_d_weight"]), rtol=4 * tol)
            np.testing.assert_allclose(float(loss.detach()), float(g[f"gen_{tag}_loss"]), rtol=tol)
            conv.weight.grad = None
            conv.bias.grad = None
            dfeat = conv.bwd(K.nchw_to_nhwc_pad(xrec.grad, K.vec(dtype) * -(-3 // K.vec(dtype)), dtype), tape)
            # gradients: relative Frobenius error (isolated ReLU / max-pool decisions flip on rounding-level pre-activations)
            assert _rel(dfeat.float().permute(0, 3, 1, 2).cpu().numpy(), g[f"gen_{tag}_dfeat"], True) < gtol
            assert _rel(conv.weight.grad.cpu().numpy(), g[f"gen_{tag}_dw"], True) < gtol
            assert _rel(conv.bias.grad.cpu().numpy(), g[f"gen_{tag}_db"], True) < gtol
            # the generator branch must not touch discriminator / VGG gradients
            assert all(p.grad is None or float(p.grad.abs().sum()) == 0 for p in loss_mod.discriminator.parameters())
            if tag != "free":
                continue
            d_loss, dlog = loss_mod(qloss, x, xrec.detach(), 1, 0, last_layer=conv.weight, split="train")
            d_loss.backward()
            np.testing.assert_allclose(float(d_loss), float(g["disc_loss"]), rtol=tol)
            np.testing.assert_allclose(float(dlog["train_logits_real"]), float(g["disc_logits_real"]), rtol=4 * tol, atol=tol * 0.1)
            np.testing.assert_allclose(float(dlog["train_logits_fake"]), float(g["disc_logits_fake"]), rtol=4 * tol, atol=tol * 0.1)
            for n, p in loss_mod.discriminator.named_parameters():
                ref = g["disc_d." + n]
                if float(np.abs(ref).max()) == 0:
                    assert float(p.grad.abs().max()) < 1e-6
                else:
                    assert _rel(p.grad.cpu().numpy(), ref, True) < gtol, n
            for n, b in loss_mod.discriminator.named_buffers():
                ref = g["disc_buf." + n]
                if ref.ndim == 0:
                    assert int(b) == int(ref)
                else:
                    np.testing.assert_allclose(b.cpu().numpy(), ref, rtol=max(tol, 1e-4), atol=tol * 1e-2)


def test_full_objective_train_step(dev):
    """the reference's complete two-optimizer step (L1 + LPIPS + adaptive GAN + codebook; then the discriminator) on the
    small DQ-VAE in bf16: finite losses, both parameter sets move, and the reuse-forward mode gives the same generator loss"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.trainer import Trainer
    from test_gpu_model import build
    with rt.compute_dtype_ctx(torch.bfloat16):
        model, _ = build("small", dev, "spread", loss="full")
        model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-4, 100, 10
        model.train()
        x = torch.from_numpy(synth.half_flat_images(4, 64, seed=99)).to(dev)
        tr = Trainer(model, max_steps=2)
        assert len(tr.opts) == 2
        w0 = model.decoder.conv_out.weight.detach().clone()
        d0 = model.loss.discriminator.main[0].weight.detach().clone()
        v0 = model.loss.perceptual_loss.net.slice1[0].weight.detach().clone()
        l0 = tr.train_step({"image": x}, 0)
        l1 = tr.train_step({"image": x}, 1)
        assert len(l0) == 2 and all(torch.isfinite(l).all() for l in l0 + l1)
        assert not torch.equal(w0, model.decoder.conv_out.weight.detach())
        assert not torch.equal(d0, model.loss.discriminator.main[0].weight.detach())
        assert torch.equal(v0, model.loss.perceptual_loss.net.slice1[0].weight.detach())      # VGG16 is frozen
        assert 0.0 <= float(model._logged["train_d_weight"]) <= 0.75
        assert int(model.loss.discriminator.main[3].num_batches_tracked) == 6                  # 3 D passes per step


def test_adaptive_weight_bf16_vs_fp32_full_width(dev):
    """bf16 is the precision the headline number is measured in: on the full-width model (BASELINE config 1 geometry: ch 128,
    K = 1024, D = 256, PatchGAN ndf 64, 64x64 images) the generator branch's adaptive GAN weight -- a RATIO of two last-layer
    gradient norms -- and both norms must agree between the bf16 and the fp32 instantiation of the same HIP kernels"""
    from dynamicvectorquantization_amd import runtime as rt
    from test_gpu_model import _report, build
    x = torch.from_numpy(synth.half_flat_images(4, 64, seed=77)).to(dev)
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        with rt.compute_dtype_ctx(dtype):
            torch.manual_seed(0)
            model, _ = build("c1", dev, "spread", loss="full")
            model.loss.disc_weight_max = None                 # the unclamped ratio is the sensitive quantity
            model.train()
            loss = model.training_step({"image": x}, 0, 0)
            n_nll, n_g = model.loss.last_adaptive_norms
            res[dtype] = dict(d_weight=float(model._logged["train_d_weight"]), nll=float(n_nll), g=float(n_g),
                              loss=float(loss.detach()), p=float(model._logged["train_p_loss"]), g_loss=float(model._logged["train_g_loss"]))
    a, b = res[torch.float32], res[torch.bfloat16]
    rel = {k: abs(b[k] - a[k]) / max(1e-12, abs(a[k])) for k in a}
    _report("adaptive_weight_bf16_vs_fp32", fp32=a, bf16=b, rel=rel)
    # measured on MI355X: nll-gradient norm 0.35 %, GAN-gradient norm 5.6 %, d_weight 5.0 %, loss 0.75 %.
    # The GAN branch is the sensitive one: d(-mean D(xrec)) / d logits is a CONSTANT map, so after the last conv the gradient is
    # nearly constant per channel and BatchNorm's backward subtracts its mean -- what is left is a small residual of values that
    # were stored as bf16.  That residual is rounding noise: three bit-different but equally exact builds of the same kernels (bias
    # added after / before the reduction, GroupNorm statistics per channel / per channel pair) gave 5.6 %, 12.7 % and 15.1 % for the
    # GAN-gradient norm at an unchanged 0.13 - 0.35 % for the nll-gradient norm.  Bounds: 1.5x the largest of them for the two
    # noise-dominated numbers, 1.5x the measurement elsewhere.
    # Round 6: two more equally exact realisations of the GroupNorm statistics -- the vector path of the current build gives nll 0.47 %,
    # p 0.41 %, g_loss 1.43 %; the statistics on the matrix pipe (DVQ_HALO_MFMA_STATS, the default) nll 0.72 %, p 0.04 %, g_loss 0.10 %
    # (same box, repeated): the nll-gradient norm moves within the same rounding noise, so its bound is 1.5x the largest seen as well.
    assert rel["nll"] <= 1.1e-2 and rel["g"] <= 0.23 and rel["d_weight"] <= 0.2, (rel, a, b)
    # (the generator loss contains d_weight * g_loss, ~0.22 of 1.07: it follows the adaptive weight's noise at a fifth of its size)
    assert rel["loss"] <= 0.25 * 0.2 and rel["p"] <= 6e-3, (rel, a, b)
