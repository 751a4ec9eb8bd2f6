"""GPU parity of the assembled DQ-VAE (encoder -> VQ -> decoder) against reference goldens, plus a
bf16 train-step smoke.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from dynamicvectorquantization_amd import synth
from test_oracle_golden import DQVAE_CFG, dqvae_state_dict

pytestmark = pytest.mark.gpu


def model_config(ch, resolution, latent, zc, k, attn_enc, attn_dec, loss="dummy", ndf=16):
    lossconfig = {"target": "modules.losses.vqperceptual.DummyLoss"}
    if loss == "ae":
        lossconfig = {"target": "modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator", "params": dict(
            disc_start=0, disc_config={"target": "modules.discriminator.model.NLayerDiscriminator",
                                       "params": dict(input_nc=3, ndf=64, n_layers=3, use_actnorm=False)},
            disc_init=True, codebook_weight=1.0, pixelloss_weight=1.0, disc_factor=0.0, disc_weight=1.0,
            perceptual_weight=0.0, disc_conditional=False, disc_loss="hinge", disc_weight_max=0.75)}
    if loss == "full":      # the shipped objective: L1 + LPIPS + adaptive hinge GAN + codebook (configs/stage1/*.yml)
        lossconfig = {"target": "modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator", "params": dict(
            disc_start=0, disc_config={"target": "modules.discriminator.model.NLayerDiscriminator",
                                       "params": dict(input_nc=3, ndf=ndf, n_layers=3, use_actnorm=False)},
            disc_init=True, codebook_weight=1.0, pixelloss_weight=1.0, disc_factor=1.0, disc_weight=1.0,
            perceptual_weight=1.0, disc_conditional=False, disc_loss="hinge", disc_weight_max=0.75)}
    return {"target": "models.stage1_dynamic.dqvae_dual_entropy.DualGrainVQModel", "params": dict(
        encoderconfig={"target": "modules.dynamic_modules.EncoderDual.DualGrainEncoder", "params": dict(
            ch=ch, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=attn_enc, dropout=0.0,
            resamp_with_conv=True, in_channels=3, resolution=resolution, z_channels=zc, update_router=False,
            router_config={"target": "modules.dynamic_modules.RouterDual.DualGrainFixedEntropyRouter", "params": dict(
                json_path=os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json"),
                fine_grain_ratito=0.5)})},
        decoderconfig={"target": "modules.dynamic_modules.DecoderPositional.Decoder", "params": dict(
            ch=ch, in_ch=zc, out_ch=3, ch_mult=[1, 1, 2, 2], num_res_blocks=2, resolution=resolution,
            attn_resolutions=attn_dec, latent_size=latent, window_size=2, position_type="fourier+learned")},
        lossconfig=lossconfig,
        vqconfig={"target": "modules.vector_quantization.quantize2_mask.VectorQuantize2", "params": dict(
            codebook_size=k, codebook_dim=zc, channel_last=False, accept_image_fmap=True, commitment_beta=0.25,
            decay=0.99, restart_unused_codes=True)},
        quant_before_dim=zc, quant_after_dim=zc, quant_sample_temperature=0.0, image_key="image",
        image_size=resolution)}


# code indices that may differ from the reference's at ITS OWN initialisation (codebook U(+-1/K): every score gap is at fp32
# rounding level, so both the reference's fp32 argmin and 1e-6 activation differences flip near ties).  Bounds = 2x the counts
# measured on MI355X (DESIGN.md section 5 lists them); every differing row must ALSO have a recorded exact gap < 1e-4.
REFINIT_MISMATCH_BOUND = {"small": 1, "c1": 2}       # measured: 0 / 128 (shrunken model), 1 / 128 with gap 6.1e-8 (full width)


def _report(kind, **kw):
    """append a measurement to gpurun_out/test_reports.jsonl when that scratch directory exists (GPU box runs)"""
    import json
    d = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_reports.jsonl"), "a") as f:
            f.write(json.dumps(dict(kind=kind, **kw)) + "\n")


GEOM = {"small": dict(ch=32, resolution=64, latent=8, zc=64, k=512, attn_enc=[4, 8], attn_dec=[8]),
        "c1": dict(ch=128, resolution=64, latent=8, zc=256, k=1024, attn_enc=[4, 8], attn_dec=[8])}


def build(tag, dev, variant, loss="dummy"):
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden(f"dqvae_{tag}")
    model = instantiate_from_config(model_config(**GEOM[tag], loss=loss)).to(dev)
    sd = dqvae_state_dict(g, variant, DQVAE_CFG[tag]["k"], DQVAE_CFG[tag]["zc"])
    own = model.state_dict()
    ae_keys = [k for k in own if not k.startswith("loss.")]
    assert sorted(ae_keys) == sorted(sd.keys()), set(ae_keys) ^ set(sd.keys())
    for k in ae_keys:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=False)
    return model, g


@pytest.mark.parametrize("impl", [2, 1])
@pytest.mark.parametrize("tag", ["small", "c1"])
def test_dqvae_forward_backward_golden(dev, tag, impl):
    from dynamicvectorquantization_amd import runtime as rt
    if tag == "c1" and impl == 1:
        pytest.skip("naive kernels on the full-width model are slow; covered by `small`")
    x = torch.from_numpy(synth.half_flat_images(2, 64, seed=4321)).to(dev)
    with rt.compute_dtype_ctx(torch.float32), rt.impl_ctx(impl):
        for variant in ("spread", "refinit"):
            model, g = build(tag, dev, variant)
            model.eval()
            rec, qloss, grain, gate, ent = model(x)
            assert np.array_equal(grain.cpu().numpy().astype(np.int8), g[f"{variant}_grain"])
            np.testing.assert_allclose(ent.cpu().numpy(), g[f"{variant}_entropy"], rtol=2e-5)
            codes = model._last["codes"].cpu().numpy().astype(np.int32)
            ref_codes = g[f"{variant}_codes"]
            bad = np.nonzero(codes.reshape(-1) != ref_codes.reshape(-1))[0]
            if variant == "spread":
                assert len(bad) == 0, f"{len(bad)} code indices differ"
                np.testing.assert_allclose(rec.detach().cpu().numpy(), g[f"{variant}_rec"], rtol=1e-3, atol=1e-3)
                np.testing.assert_allclose(qloss.item(), g[f"{variant}_qloss"], rtol=1e-3)
                gout = torch.from_numpy(synth.det_param(f"dqvae.{tag}.gout", tuple(rec.shape))).to(dev)
                ((rec * gout).sum() / rec.numel() * 100.0 + qloss).backward()
                params = dict(model.named_parameters())
                for key in [k for k in g.files if k.startswith("grad.")]:
                    name = key[5:]
                    got = params[name].grad.cpu().numpy()
                    ref = g[key]
                    if got.size != ref.size:
                        got = got.reshape(-1)[:: max(1, got.size // 20000)]
                    s = max(1e-9, float(np.abs(ref).max()))
                    err = float(np.abs(got.reshape(ref.shape) - ref).max()) / s
                    assert err < 5e-3, f"{name}: rel-to-max grad error {err}"
            else:
                # reference-init codebook U(+-1/K): near ties.  Differences are only allowed on rows whose
                # exact top-2 gap (recorded from the reference's own activations) is at fp32-noise level
                _report("refinit_code_mismatch", tag=tag, impl=impl, mismatched=int(len(bad)), total=int(codes.size),
                        max_gap=float(g[f"{variant}_gap"][bad].max()) if len(bad) else 0.0)
                assert len(bad) <= REFINIT_MISMATCH_BOUND[tag], f"{len(bad)} of {codes.size} code indices differ from the reference"
                assert np.all(g[f"{variant}_gap"][bad] < 1e-6), g[f"{variant}_gap"][bad]


def test_dqvae_forward_golden_fp32x3(dev):
    """the reference goldens in `fp32x3` (fp32 tensors, matrix products as three bf16 MFMA passes on split operands): the fast mode that is
    meant to meet north_star's tolerance -- grain map exact, code indices exact on the spread codebook, reconstruction within 1e-3"""
    from dynamicvectorquantization_amd import runtime as rt
    x = torch.from_numpy(synth.half_flat_images(2, 64, seed=4321)).to(dev)
    for tag in ("small", "c1"):
        with rt.compute_dtype_ctx("fp32x3"):
            model, g = build(tag, dev, "spread")
            model.eval()
            rec, qloss, grain, gate, ent = model(x)
            assert np.array_equal(grain.cpu().numpy().astype(np.int8), g["spread_grain"])
            codes = model._last["codes"].cpu().numpy().astype(np.int32)
            assert np.array_equal(codes.reshape(-1), g["spread_codes"].reshape(-1)), tag
            np.testing.assert_allclose(rec.detach().cpu().numpy(), g["spread_rec"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(qloss.item(), g["spread_qloss"], rtol=1e-3)
            err = float(np.linalg.norm(rec.detach().cpu().numpy() - g["spread_rec"]) / np.linalg.norm(g["spread_rec"]))
            _report("fp32x3_recon_rel_err", tag=tag, err=err)
            assert err < 1e-3, (tag, err)
            # the backward of the same mode against the reference's parameter gradients (the `grad.*` entries the fp32 test uses)
            gout = torch.from_numpy(synth.det_param(f"dqvae.{tag}.gout", tuple(rec.shape))).to(dev)
            ((rec * gout).sum() / rec.numel() * 100.0 + qloss).backward()
            params = dict(model.named_parameters())
            worst = 0.0
            for key in [k for k in g.files if k.startswith("grad.")]:
                got = params[key[5:]].grad.cpu().numpy()
                ref = g[key]
                if got.size != ref.size:
                    got = got.reshape(-1)[:: max(1, got.size // 20000)]
                e = float(np.abs(got.reshape(ref.shape) - ref).max()) / max(1e-9, float(np.abs(ref).max()))
                worst = max(worst, e)
                assert e < 5e-3, f"{tag} {key}: rel-to-max grad error {e}"
            _report("fp32x3_grad_rel_to_max", tag=tag, worst=worst)
    assert not rt.fp32_split()


def test_train_step_bf16_smoke(dev):
    """two optimizer steps of the AE-only objective in bf16: finite loss that moves, EMA buffers updated"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.trainer import Trainer
    with rt.compute_dtype_ctx(torch.bfloat16):
        model, _ = build("small", dev, "spread", loss="ae")
        model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-4, 100, 10
        model.train()
        x = torch.from_numpy(synth.half_flat_images(4, 64, seed=99)).to(dev)
        tr = Trainer(model, max_steps=2)
        w0 = model.decoder.conv_out.weight.detach().clone()
        n0 = model.quantize.codebook.cluster_size_ema.clone()
        l0 = tr.train_step({"image": x}, 0)
        l1 = tr.train_step({"image": x}, 1)
        assert all(torch.isfinite(l).all() for l in l0 + l1)
        assert not torch.equal(w0, model.decoder.conv_out.weight.detach())
        assert not torch.equal(n0, model.quantize.codebook.cluster_size_ema)


def test_packed_weights_follow_the_optimizer(dev):
    """after optimizer steps every conv's packed copies (weights AND the zero-padded bias of convs whose Cout is not a
    multiple of 8) equal the fp32 masters -- including modules whose re-pack was triggered by another module"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    from dynamicvectorquantization_amd.trainer import Trainer
    with rt.compute_dtype_ctx(torch.bfloat16):
        model, _ = build("small", dev, "spread", loss="full")
        model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 100, 10
        model.train()
        x = torch.from_numpy(synth.half_flat_images(2, 64, seed=7)).to(dev)
        tr = Trainer(model, max_steps=2)
        b0 = model.decoder.conv_out.bias.detach().clone()
        tr.train_step({"image": x}, 0)
        tr.train_step({"image": x}, 1)
        with torch.no_grad():
            model(x)                                     # forward after the last step: packs are refreshed lazily
        assert not torch.equal(b0, model.decoder.conv_out.bias.detach())
        checked = 0
        for name, m in model.named_modules():
            if not isinstance(m, Conv2d) or torch.bfloat16 not in m._packs or name.startswith("loss.discriminator"):
                continue
            ent = m._packs[torch.bfloat16]
            w = m.weight.detach().permute(0, 2, 3, 1).to(torch.bfloat16)
            assert torch.equal(ent["w"][: m.out_channels, :, :, : m.in_channels], w), name
            if ent["bias"] is not None:
                assert torch.equal(ent["bias"][: m.out_channels], m.bias.detach()), name
                checked += 1
        assert checked >= 1


def test_trainer_validation_loop_and_monitor_checkpoints(dev, tmp_path):
    """Lightning's validation loop + ModelCheckpoint(monitor) of the reference's train.py:152-183 / dqvae_dual_entropy.py:185-201:
    Trainer.validate runs validation_step in eval mode (no parameter, EMA-buffer or BatchNorm-statistic moves), averages the logged
    scalars, restores train mode; fit() validates every `val_every` steps and keeps the `save_top_k` best checkpoints by `monitor`"""
    import os
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.trainer import Trainer
    with rt.compute_dtype_ctx(torch.bfloat16):
        torch.manual_seed(0)
        model, _ = build("small", dev, "spread", loss="full")
        model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 100, 2
        model.monitor = "val_rec_loss"
        model.train()
        tr = Trainer(model, max_steps=6, use_graph=False)
        xs = [torch.from_numpy(synth.half_flat_images(2, 64, seed=40 + i)).to(dev) for i in range(4)]
        vs = [{"image": torch.from_numpy(synth.half_flat_images(2, 64, seed=90 + i)).to(dev)} for i in range(3)]
        tr.train_step({"image": xs[0]}, 0)
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        m1 = tr.validate(vs)
        assert model.training, "train mode must be restored"
        after = model.state_dict()
        assert all(torch.equal(before[k], after[k]) for k in before), "validation must not move any parameter or buffer"
        for key in ("val_rec_loss", "val_aeloss", "val_total_loss", "val_quant_loss", "val_disc_loss", "val_fine_ratio"):
            assert key in m1 and np.isfinite(m1[key]), (key, sorted(m1))
        assert m1["val_d_weight"] == 0.0                         # reference: the adaptive weight's autograd.grad fails in eval -> 0
        m2 = tr.validate(vs)
        assert abs(m1["val_rec_loss"] - m2["val_rec_loss"]) <= 1e-5 * abs(m1["val_rec_loss"])          # eval forward: repeatable
        assert abs(m1["val_rec_loss"] - tr.validate(vs[:1])["val_rec_loss"]) > 0                       # a mean over the batches given
        ckpt = str(tmp_path / "checkpoints" / "last.ckpt")
        tr.fit(lambda step: {"image": xs[step % 4]}, ckpt_path=ckpt, val_fn=lambda: iter(vs), val_every=2, save_top_k=2)
        files = sorted(os.listdir(os.path.dirname(ckpt)))
        assert "last.ckpt" in files
        best = [f for f in files if f.startswith("epoch=") and "val_rec_loss=" in f]
        assert 1 <= len(best) <= 2, files                        # three validations (steps 2, 4, 6), at most save_top_k kept
        sd = torch.load(os.path.join(os.path.dirname(ckpt), best[0]), map_location="cpu", weights_only=False)
        assert "state_dict" in sd and "optimizer_states" in sd and np.isfinite(tr.last_val_metrics["val_rec_loss"])


def test_trainer_checkpoint_resume(dev):
    """Trainer.state_dict / load_state_dict (train.py -r): a fresh process state restored from the checkpoint continues the run --
    same parameters, Adam moments, LR-schedule position, step counter, VQ-EMA and BatchNorm buffers -> same next losses"""
    import copy
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.trainer import Trainer

    def make(seed):
        torch.manual_seed(seed)
        model, _ = build("small", dev, "spread", loss="full")
        model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 100, 10
        model.train()
        return model, Trainer(model, max_steps=8)

    xs = [torch.from_numpy(synth.half_flat_images(2, 64, seed=20 + i)).to(dev) for i in range(4)]
    with rt.compute_dtype_ctx(torch.bfloat16):
        model, tr = make(0)
        for i in range(2):
            tr.train_step({"image": xs[i]}, i)
        ckpt = copy.deepcopy(tr.state_dict())
        assert ckpt["global_step"] == 2 and len(ckpt["optimizer_states"]) == 2
        # optimizer states are written in torch's own (= Lightning's) format: they load into a plain torch.optim.Adam built
        # over the same parameter list, i.e. the reference can resume from this file
        os0 = ckpt["optimizer_states"][0]
        assert float(os0["state"][0]["step"]) == 2 and os0["param_groups"][0]["betas"] == (0.5, 0.9)
        probe = torch.optim.Adam([torch.nn.Parameter(torch.empty_like(p_)) for p_ in model.ae_parameters()], lr=1e-3, betas=(0.5, 0.9))
        probe.load_state_dict(copy.deepcopy(os0))
        n_state = sum(1 for p_ in model.ae_parameters() if p_.requires_grad)
        assert len(os0["state"]) == n_state and tuple(os0["state"][0]["exp_avg"].shape) == tuple(model.ae_parameters()[0].shape)
        ref_losses = [[float(l) for l in tr.train_step({"image": xs[i]}, i)] for i in (2, 3)]
        ref_w = model.decoder.conv_out.weight.detach().float().cpu().clone()
        ref_lr = [g["lr"] for o in tr.opts for g in o.param_groups]

        model2, tr2 = make(123)                                   # different init: everything must come from the checkpoint
        with torch.no_grad():
            for p_ in model2.parameters():
                p_.add_(0.01)
        tr2.load_state_dict(ckpt)
        assert model2.global_step == 2
        # restored state is bit-identical to the checkpoint: parameters + buffers (through the flat-buffer views), Adam
        # moments and step counts, LambdaLR position
        sd2 = model2.state_dict()
        assert list(sd2.keys()) == list(ckpt["state_dict"].keys())
        for k_, v_ in ckpt["state_dict"].items():
            assert torch.equal(sd2[k_].detach().cpu(), v_.detach().cpu()), k_
        for o2, o1, st in zip(tr2.opts, tr.opts, ckpt["optimizer_states"]):
            assert o2._fstate["step"] == 2
            back = tr2._optimizer_state_dict(o2)
            assert sorted(back["state"].keys()) == sorted(st["state"].keys())
            for i_, ps in st["state"].items():
                assert torch.equal(back["state"][i_]["exp_avg"], ps["exp_avg"]) and torch.equal(back["state"][i_]["exp_avg_sq"], ps["exp_avg_sq"])
        # the round-1 flat format still loads; a mismatching parameter set is refused instead of silently truncated
        legacy = {"step": 2, "exp_avg": tr.opts[0]._fstate["m"].cpu() * 0 + 1.0, "exp_avg_sq": tr.opts[0]._fstate["v"].cpu() * 0 + 2.0}
        tr2._load_optimizer_state(tr2.opts[0], legacy)
        assert float(tr2.opts[0]._fstate["m"].mean()) == 1.0 and float(tr2.opts[0]._fstate["v"].mean()) == 2.0
        with pytest.raises(ValueError):
            tr2._load_optimizer_state(tr2.opts[0], {"step": 2, "exp_avg": torch.zeros(5), "exp_avg_sq": torch.zeros(5)})
        tr2.load_state_dict(ckpt)
        got_lr_now = [g["lr"] for o in tr2.opts for g in o.param_groups]
        got_losses = [[float(l) for l in tr2.train_step({"image": xs[i]}, i)] for i in (2, 3)]
        got_lr = [g["lr"] for o in tr2.opts for g in o.param_groups]
    assert got_lr == ref_lr and all(a > 0 for a in got_lr_now)
    # the continued run follows the original one (bf16 + atomics + the adaptive GAN weight: loose bound, the exact part is above)
    np.testing.assert_allclose(np.array(got_losses), np.array(ref_losses), rtol=0.25, atol=0.05)
