"""GPU parity of the stage-2 input permutation (integer work: bit-exact) against the reference goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

KEYS = ("coarse_content", "fine_content", "coarse_position", "fine_position", "coarse_segment", "fine_segment")


@pytest.mark.parametrize("order", ["region-first", "row-first"])
@pytest.mark.parametrize("name,hw1", [("small", 4), ("full", 16)])
def test_permuter_golden(dev, order, name, hw1):
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden("permuter")
    tag = order.split("-")[0]
    perm = instantiate_from_config({"target": "modules.dynamic_modules.permuter.DualGrainSeperatePermuter",
                                    "params": dict(coarse_hw=hw1, fine_hw=2 * hw1, fine_position_order=order)})
    idx = torch.from_numpy(g[f"{tag}_{name}_indices"]).to(dev)
    grain = torch.from_numpy(g[f"{tag}_{name}_grain"]).to(dev)
    out = perm(idx, grain)
    for k in KEYS:
        assert out[k].dtype == torch.long
        assert np.array_equal(out[k].cpu().numpy(), g[f"{tag}_{name}_{k}"]), k
    back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
    assert np.array_equal(back.cpu().numpy(), g[f"{tag}_{name}_back"])


@pytest.mark.parametrize("order", ["region-first", "row-first"])
def test_permuter_reference_known_answer_vector(dev, order):
    """the reference's only in-repo known-answer vector (modules/dynamic_modules/permuter.py:181-285, `test_code == 2`: two 32x32
    code maps + 16x16 grain maps with the reference's code / position constants): forward equals what the reference's permuter
    emits for it, forward -> forward_back reproduces the code maps (the self-test prints True)"""
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden("permuter")
    tag = order.split("-")[0]
    perm = instantiate_from_config({"target": "modules.dynamic_modules.permuter.DualGrainSeperatePermuter", "params": dict(
        coarse_hw=16, fine_hw=32, content_pad_code=1024, content_eos_code=1025, coarse_position_pad_code=256,
        coarse_position_eos_code=257, fine_position_pad_code=1024, fine_position_eos_code=1025, fine_position_order=order)})
    idx = torch.from_numpy(g["known_indices"]).to(dev)
    out = perm(idx, torch.from_numpy(g["known_grain"]).to(dev))
    for k in KEYS:
        assert np.array_equal(out[k].cpu().numpy(), g[f"known_{tag}_{k}"]), k
    back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
    assert np.array_equal(back.cpu().numpy(), g[f"known_{tag}_back"])
    assert bool(torch.all(back == idx))


def test_permuter_malformed_sequences_golden(dev):
    """missing EOS, early EOS, duplicate positions: the reference's sequential semantics"""
    from dynamicvectorquantization_amd.stage2 import DualGrainSeperatePermuter
    g = load_golden("permuter")
    perm = DualGrainSeperatePermuter(coarse_hw=4, fine_hw=8)
    back = perm.forward_back(*[torch.from_numpy(g[k]).to(dev) for k in ("mal_cc", "mal_fc", "mal_cp", "mal_fp")])
    assert np.array_equal(back.cpu().numpy(), g["mal_back"])


@pytest.mark.parametrize("order", ["region-first", "row-first"])
def test_permuter_full_size_round_trip_and_oracle(dev, order):
    """BASELINE geometry (16x16 coarse cells, 32x32 codes), batch 64: encode -> decode reproduces every code map whose coarse
    cells hold one repeated code (the property the reference's own self-test prints), and the rows equal the oracle's"""
    from dynamicvectorquantization_amd.stage2 import DualGrainSeperatePermuter
    from oracle import permuter as ope
    rs = np.random.RandomState(7)
    b = 64
    fine = rs.randint(0, 1024, size=(b, 32, 32)).astype(np.int64)
    grain = (rs.uniform(size=(b, 16, 16)) < rs.uniform(size=(b, 1, 1))).astype(np.int64)
    rep = grain.repeat(2, axis=1).repeat(2, axis=2)
    coarse = fine[:, ::2, ::2].repeat(2, axis=1).repeat(2, axis=2)
    codes = np.where(rep == 1, fine, coarse)
    perm = DualGrainSeperatePermuter(fine_position_order=order)
    out = perm(torch.from_numpy(codes).to(dev), torch.from_numpy(grain).to(dev))
    ora = ope.forward(codes, grain, 16, 2, order)
    for k in KEYS:
        assert np.array_equal(out[k].cpu().numpy(), ora[k]), k
    back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
    assert np.array_equal(back.cpu().numpy(), codes)
    # sequence lengths: one coarse token per coarse cell, four per fine cell, plus EOS
    n_fine = grain.reshape(b, -1).sum(1)
    assert out["coarse_content"].shape[1] == int((256 - n_fine).max()) + 1
    assert out["fine_content"].shape[1] == int(4 * n_fine.max()) + 1


def test_sos_providers(dev):
    from dynamicvectorquantization_amd.config import instantiate_from_config
    p = instantiate_from_config({"target": "modules.dynamic_modules.label_provider.PositionAwareSOSProvider",
                                 "params": dict(coarse_sos=1026, coarse_pos_sos=258, fine_sos=1026, fine_pos_sos=1026,
                                                coarse_seg_sos=0, fine_seg_sos=1)})
    x = torch.zeros(3, 5, device=dev)
    outs = p.encode(x)
    assert [int(o[0, 0]) for o in outs] == [1026, 1026, 258, 1026, 0, 1] and all(o.shape == (3, 1) and o.dtype == torch.long for o in outs)
    c = instantiate_from_config({"target": "modules.dynamic_modules.label_provider.ClassForContentOnlyPositionAwareSOSProvider",
                                 "params": dict(n_classes=1000, threshold=1026, coarse_pos_sos=258, fine_pos_sos=1026)})
    lab = torch.tensor([3, 999], device=dev)
    o = c.encode(lab)
    assert o[0].tolist() == [[1029], [2025]] and o[1].tolist() == [[1029], [2025]] and o[4] is None


STACKGPT_CFG = dict(vocab_size=1027, coarse_position_size=259, fine_position_size=1027, segment_size=2, block_size=64,
                    position_layer=2, content_layer=3, n_head=4, n_embd=64, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0,
                    content_pad_code=1024, coarse_position_pad_code=256, fine_position_pad_code=1024, activate_pad_ignore=True)


def stackgpt_inputs(seed=3, b=3):
    """the fixture's ragged teacher-forcing batch (same generator as tools/gen_golden.py)"""
    rs = np.random.RandomState(seed)
    n_c, n_f = [6, 9, 3], [16, 8, 20]
    lc, lf = max(n_c) + 2, max(n_f) + 2
    cc = np.full((b, lc), 1024, dtype=np.int64); cp = np.full((b, lc), 256, dtype=np.int64)
    fc = np.full((b, lf), 1024, dtype=np.int64); fp = np.full((b, lf), 1024, dtype=np.int64)
    for i in range(b):
        cc[i, 0], cp[i, 0], fc[i, 0], fp[i, 0] = 1026, 258, 1026, 1026
        cc[i, 1:1 + n_c[i]] = rs.randint(0, 1024, n_c[i]); cc[i, 1 + n_c[i]] = 1025
        cp[i, 1:1 + n_c[i]] = np.sort(rs.choice(256, n_c[i], replace=False)); cp[i, 1 + n_c[i]] = 257
        fc[i, 1:1 + n_f[i]] = rs.randint(0, 1024, n_f[i]); fc[i, 1 + n_f[i]] = 1025
        fp[i, 1:1 + n_f[i]] = np.sort(rs.choice(1024, n_f[i], replace=False)); fp[i, 1 + n_f[i]] = 1025
    cs, fs = np.zeros_like(cc), np.ones_like(fc)
    return dict(coarse_content=cc, fine_content=fc, coarse_position=cp, fine_position=fp, coarse_seg=cs, fine_seg=fs,
                content_target=np.concatenate([cc, fc], 1)[:, 1:], coarse_position_target=cp[:, 1:], fine_position_target=fp)


def build_stackgpt(dev):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    model = instantiate_from_config({"target": "modules.dynamic_modules.stackgpt.StackGPT", "params": STACKGPT_CFG}).to(dev)
    with torch.no_grad():
        for n, p in model.named_parameters():
            v = synth.det_param("stackgpt." + n, tuple(p.shape))
            p.copy_(torch.from_numpy(v * (0.3 if n == "pos_emb" else 1.0)).to(dev))
    return model


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 5e-3), ("fp32x3", 3e-4, 6e-3), (torch.bfloat16, 2e-2, 8e-2)])
def test_stackgpt_golden(dev, dtype, tol, gtol):
    """teacher-forced losses, parameter gradients (incl. embeddings with padding rows) and logits vs the reference StackGPT"""
    from dynamicvectorquantization_amd import runtime as rt
    g = load_golden("stackgpt")
    with rt.compute_dtype_ctx(dtype):
        model = build_stackgpt(dev).train()
        own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ref = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(g["state_keys"], g["state_shapes"])}
        assert own == ref, set(own) ^ set(ref)
        inp = {k: torch.from_numpy(v).to(dev) for k, v in stackgpt_inputs().items()}
        out = model(**inp)
        for k in ("position_loss", "content_loss", "coarse_position_loss", "fine_position_loss"):
            np.testing.assert_allclose(float(out[k].detach()), float(g[k]), rtol=tol)
        (1.0 * out["content_loss"] + 0.7 * out["position_loss"]).backward()
        params = dict(model.named_parameters())
        for key in [k for k in g.files if k.startswith("grad.")]:
            ref_g = g[key].astype(np.float64)
            got = params[key[5:]].grad.cpu().numpy().astype(np.float64).reshape(ref_g.shape)
            err = float(np.linalg.norm(got - ref_g)) / max(1e-30, float(np.linalg.norm(ref_g)))
            assert err < gtol, f"{key}: relative Frobenius error {err}"
            if key == "grad.content_emb.weight":       # the padding row never receives gradient (nn.Embedding(padding_idx))
                assert float(np.abs(got[1024]).max()) == 0.0
        model.eval()
        lo = model(**{k: v for k, v in inp.items() if not k.endswith("target")})
        for k in ("position_logits", "content_logits"):
            assert tuple(lo[k].shape) == g[k].shape
            ref_l = g[k]
            assert float(np.abs(lo[k].cpu().numpy() - ref_l).max()) < (6e-2 if dtype == torch.bfloat16 else 2e-3) * float(np.abs(ref_l).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_dropout_backward_fused_into_layernorm_backward(dev, dtype, monkeypatch):
    """with dropout on, the backward of the two per-block dropouts is written by the LayerNorm backward that produces their operand
    (dvq_layernorm_bwd_res_drop): same seeds -> the same masks as the stand-alone dvq_dropout passes, so every parameter gradient equals
    the unfused run's (up to the summation order of the atomically accumulated weight gradients)"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    # kernel level: second output == dvq_dropout of the first, bit for bit
    rows, c = 333, 1024
    x = torch.randn(rows, c, device=dev).to(dtype)
    dy = torch.randn(rows, c, device=dev).to(dtype)
    res = torch.randn(rows, c, device=dev).to(dtype)
    gamma = torch.rand(c, device=dev) + 0.5
    mr = torch.stack([x.float().mean(1), 1.0 / (x.float().var(1, unbiased=False) + 1e-5).sqrt()], 1).contiguous()
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    dx0 = K.layernorm_bwd(x, dy, mr, gamma, dg.clone(), db.clone(), res)
    dx1, dxd = K.layernorm_bwd(x, dy, mr, gamma, dg.clone(), db.clone(), res, drop=(0.1, 987654321))
    assert torch.equal(dx0, dx1) and torch.equal(dxd, K.dropout(dx1, 0.1, 987654321))
    frac = float((dxd == 0).float().mean())
    assert 0.08 < frac < 0.12, frac
    # model level
    grads = {}
    with rt.compute_dtype_ctx(dtype):
        for mode in ("1", "0"):
            monkeypatch.setenv("DVQ_FUSE_DROP_BWD", mode)
            torch.manual_seed(0)
            rt._seed_counter[0] = 0
            model = build_stackgpt(dev).train()
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.1
            inp = {k: torch.from_numpy(v).to(dev) for k, v in stackgpt_inputs().items()}
            out = model(**inp)
            (1.0 * out["content_loss"] + 0.7 * out["position_loss"]).backward()
            grads[mode] = ({n: p.grad.detach().clone() for n, p in model.named_parameters()}, float(out["content_loss"].detach()))
    assert abs(grads["1"][1] - grads["0"][1]) <= 1e-6 * abs(grads["0"][1])      # (the loss sums are folded with atomics)
    for n, ga in grads["1"][0].items():
        gb = grads["0"][0][n]
        assert float((ga - gb).norm()) <= 1e-4 * float(gb.norm()) + 1e-9, n


def dualformer_config():
    import os
    from conftest import REPO
    from test_gpu_model import GEOM, model_config
    fs = model_config(**GEOM["small"])                                  # frozen small DQ-VAE: 64x64 -> 8x8 codes, K = 512
    return {"target": "models.stage2_dynamic.dqtransformer_uncond_entropy.Dualformer", "params": dict(
        transformer_config={"target": "modules.dynamic_modules.stackgpt.StackGPT", "params": dict(
            vocab_size=515, coarse_position_size=19, fine_position_size=67, segment_size=2, block_size=96, position_layer=2,
            content_layer=2, n_head=4, n_embd=64, embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1, content_pad_code=512,
            coarse_position_pad_code=16, fine_position_pad_code=64, activate_pad_ignore=True)},
        first_stage_config=fs,
        uncond_stage_config={"target": "modules.dynamic_modules.label_provider.PositionAwareSOSProvider", "params": dict(
            coarse_sos=514, coarse_pos_sos=18, fine_sos=514, fine_pos_sos=66, coarse_seg_sos=0, fine_seg_sos=1)},
        permuter_config={"target": "modules.dynamic_modules.permuter.DualGrainSeperatePermuter", "params": dict(
            coarse_hw=4, fine_hw=8, content_pad_code=512, content_eos_code=513, coarse_position_pad_code=16,
            coarse_position_eos_code=17, fine_position_pad_code=64, fine_position_eos_code=65, fine_position_order="region-first")},
        weight_decay=0.01, warmup_epochs=0)}


@pytest.mark.parametrize("kind", ["uncond", "class"])
@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 3e-4, 6e-3), ("fp32x3", 4e-4, 8e-3), (torch.bfloat16, 3e-2, 1e-1)])
def test_dualformer_forward_golden(dev, kind, dtype, tol, gtol):
    """Dualformer.training_step of BOTH stage-2 models (uncond: dqtransformer_uncond_entropy.py:180-234, class-conditional:
    dqtransformer_class2_entropy.py) against the reference on a ragged image batch: frozen DQ-VAE -> codes -> permuter ->
    start tokens -> StackGPT.  Sequences bit-exact, the four losses + training loss, transformer gradients, state_dict layout."""
    from conftest import REPO
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from golden_cfg import dualformer_cfg
    from test_oracle_golden import dqvae_state_dict
    g = load_golden("dualformer")
    thr_json = os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json")
    target = {"uncond": "models.stage2_dynamic.dqtransformer_uncond_entropy.Dualformer",
              "class": "models.stage2_dynamic.dqtransformer_class2_entropy.Dualformer"}[kind]
    with rt.compute_dtype_ctx(dtype):
        model = instantiate_from_config({"target": target, "params": dualformer_cfg(kind, thr_json)}).to(dev)
        # state_dict layout = the reference's, key by key (attention-mask buffers included)
        own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ref = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(g[f"{kind}.state_keys"], g[f"{kind}.state_shapes"])}
        assert own == ref, sorted(set(own) ^ set(ref))[:8]
        fs_sd = dqvae_state_dict(load_golden("dqvae_small"), "spread", 512, 64)
        model.first_stage_model.load_state_dict(fs_sd)
        with torch.no_grad():
            for n, p in model.transformer.named_parameters():
                v = synth.det_param(f"dualformer.{kind}." + n, tuple(p.shape))
                p.copy_(torch.from_numpy(v * (0.3 if n == "pos_emb" else 1.0)).to(dev))
        rt.bump_weights_epoch()
        model.train()
        batch = {"image": torch.from_numpy(synth.ragged_grain_images(64, seed=31)).to(dev),
                 "class_label": torch.tensor([3, 0, 9], dtype=torch.long, device=dev)}
        with torch.no_grad():
            _, z = model.encode_to_z(batch["image"])
        if dtype != torch.bfloat16:         # the frozen first stage's codes (hence every sequence) are exact in parity mode (fp32 and
                                            # fp32x3: the split products leave these margins alone)
            for k in ("coarse_content", "fine_content", "coarse_position", "fine_position", "coarse_segment", "fine_segment"):
                assert np.array_equal(z[k].cpu().numpy(), g[f"{kind}.z.{k}"]), k
        else:                               # bf16 activations may flip a few near-tie codes; positions / lengths never change
            for k in ("coarse_position", "fine_position", "coarse_segment", "fine_segment"):
                assert np.array_equal(z[k].cpu().numpy(), g[f"{kind}.z.{k}"]), k
            assert (z["fine_content"].cpu().numpy() != g[f"{kind}.z.fine_content"]).mean() < 0.05
        total = model.training_step(batch, 0)
        for k in ("content_loss", "position_loss", "coarse_position_loss", "fine_position_loss"):
            np.testing.assert_allclose(float(model._logged["train_" + k]), float(g[f"{kind}.train_{k}"]), rtol=tol)
        np.testing.assert_allclose(float(total.detach()), float(g[f"{kind}.total"]), rtol=tol)
        np.testing.assert_allclose(float(model._logged["train_loss"]), float(g[f"{kind}.total"]), rtol=tol)
        total.backward()
        params = dict(model.transformer.named_parameters())
        for key in [k for k in g.files if k.startswith(f"{kind}.grad.")]:
            ref_g = g[key].astype(np.float64)
            got = params[key[len(kind) + 6:]].grad.cpu().numpy().astype(np.float64).reshape(ref_g.shape)
            err = float(np.linalg.norm(got - ref_g)) / max(1e-30, float(np.linalg.norm(ref_g)))
            assert err < gtol, f"{key}: relative Frobenius error {err}"
        assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in model.first_stage_model.parameters())


def test_dualformer_train_steps_and_round_trip(dev):
    """stage-2 plumbing end to end in bf16: frozen DQ-VAE -> codes -> permuter -> StackGPT teacher forcing -> AdamW; the loss
    falls on a repeated batch, only transformer parameters move, and codes -> sequences -> codes -> image reproduces the
    first stage's own reconstruction"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    with rt.compute_dtype_ctx(torch.bfloat16):
        torch.manual_seed(0)
        model = instantiate_from_config(dualformer_config()).to(dev)
        model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 3e-3, 0.0, 100, 10
        model.train()
        assert not model.first_stage_model.training                      # disabled_train keeps the DQ-VAE in eval mode
        x = torch.from_numpy(synth.half_flat_images(4, 64, seed=11)).to(dev)
        tr = Trainer(model, max_steps=6)
        opt = tr.opts[0]
        assert len(opt.param_groups) == 2 and opt.param_groups[0]["weight_decay"] == 0.01 and opt.param_groups[1]["weight_decay"] == 0.0
        n_decay = sum(p.numel() for p in opt.param_groups[0]["params"])
        n_lin = sum(m.weight.numel() for m in model.transformer.modules() if m.__class__.__name__ == "Linear")
        assert n_decay == n_lin
        enc0 = model.first_stage_model.encoder.conv_in.weight.detach().clone()
        w0 = model.transformer.content_head[1].weight.detach().clone()
        losses = [float(tr.train_step({"image": x}, i)[0]) for i in range(6)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        assert torch.equal(enc0, model.first_stage_model.encoder.conv_in.weight.detach())
        assert not torch.equal(w0, model.transformer.content_head[1].weight.detach())
        model.eval()
        with torch.no_grad():
            _, z = model.encode_to_z(x)
            rec2 = model.decode_to_img(z["coarse_content"], z["fine_content"], z["coarse_position"], z["fine_position"])
            fs = model.first_stage_model
            rec1 = fs(x)[0]
            codes = fs._last["codes"].view(4, 8, 8)
            rec3 = fs.decode(fs.get_code_emb_with_depth(codes).permute(0, 3, 1, 2))
            val = model.validation_step({"image": x}, 0)
        assert torch.isfinite(val)
        # sequences -> code map is lossless: same image as decoding the code map directly (GroupNorm statistics are summed with
        # atomics, so two runs may differ in the last bf16 bit of a few activations)
        assert float((rec2 - rec3).abs().max()) <= 2e-2 * float(rec3.abs().max())
        # vs the autoencoder's own forward: its straight-through x + (x_q - x) is rounded to bf16 once more than a direct
        # codebook lookup (1-ulp differences that the untrained decoder's GroupNorms amplify): loose bound only
        assert float((rec1 - rec2).abs().max()) <= 0.1 * float(rec1.abs().max())


SAMPLER_GPT_CFG = dict(vocab_size=515, coarse_position_size=19, fine_position_size=67, segment_size=2, block_size=96,
                       position_layer=2, content_layer=2, n_head=4, n_embd=64, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0,
                       content_pad_code=512, coarse_position_pad_code=16, fine_position_pad_code=64, activate_pad_ignore=True)


def _sampler_model(dev, order):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    cfg = dualformer_config()
    cfg["params"]["transformer_config"]["params"] = dict(SAMPLER_GPT_CFG)
    cfg["params"]["permuter_config"]["params"]["fine_position_order"] = order
    model = instantiate_from_config(cfg).to(dev).eval()
    with torch.no_grad():
        for n, p in model.transformer.named_parameters():
            v = synth.det_param("sampler." + n, tuple(p.shape))
            p.copy_(torch.from_numpy(v * (0.3 if n == "pos_emb" else 4.0 if n.endswith("head.1.weight") else 1.0)).to(dev))
    return model


@pytest.mark.parametrize("kv_cache", [True, False])
@pytest.mark.parametrize("order", ["region-first", "row-first"])
def test_sampler_greedy_golden(dev, order, kv_cache):
    """Dualformer.sample_from_scratch (greedy + top-k, so no RNG) reproduces the reference's sequences token for token in both
    fine-position modes, and the decoded code maps"""
    from dynamicvectorquantization_amd import runtime as rt
    g = load_golden("sampler")
    tag = order.split("-")[0]
    with rt.compute_dtype_ctx(torch.float32):
        model = _sampler_model(dev, order)
        c = model.encode_to_c(torch.zeros(3, 3, 64, 64, device=dev))
        for fix in (False, True):
            # kv_cache=True: one new row per step through K/V caches; False: the reference's whole-prefix recomputation
            res = model.sample_from_scratch(*c, temperature=1.0, sample=False, top_k=50, top_p=None, top_k_pos=None, top_p_pos=None,
                                            process=False, fix_fine_position=fix, kv_cache=kv_cache)
            for name, r in zip(("coarse_content", "fine_content", "coarse_position", "fine_position"), res):
                assert np.array_equal(r.cpu().numpy(), g[f"{tag}_{int(fix)}_{name}"]), (fix, name)
            codes = model.permuter.forward_back(*res)
            assert np.array_equal(codes.cpu().numpy(), g[f"{tag}_{int(fix)}_codes"])
        img = model.decode_to_img(*res)
        assert tuple(img.shape) == (3, 3, 64, 64) and bool(torch.isfinite(img).all())


def test_sampler_token_step_graphs(dev):
    """K/V-cached sampling replays one captured hipGraph per (transformer, position table) for the single-row steps: the graph
    path must draw exactly the tokens of the eager path (greedy), run after run on the cached state, and must drop its
    graphs when the weights change"""
    from dynamicvectorquantization_amd import runtime as rt
    with rt.compute_dtype_ctx(torch.float32):
        model = _sampler_model(dev, "region-first")
        c = model.encode_to_c(torch.zeros(3, 3, 64, 64, device=dev))
        kw = dict(temperature=1.0, sample=False, top_k=50, top_p=None, top_k_pos=None, top_p_pos=None, process=False,
                  fix_fine_position=False, kv_cache=True)
        first = [r.cpu().numpy() for r in model.sample_from_scratch(*c, **kw)]
        st = next(iter(model._decode_states.values()))
        assert st.use_graph and any(e["graph"] is not None for e in st._steps.values()), "token steps were not captured"
        again = [r.cpu().numpy() for r in model.sample_from_scratch(*c, **kw)]          # replays only
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
        st.use_graph = False
        st._steps.clear()
        eager = [r.cpu().numpy() for r in model.sample_from_scratch(*c, **kw)]
        assert all(np.array_equal(a, b) for a, b in zip(first, eager))
        # weights change -> the captured graphs (which hold the old compute-dtype copies) are discarded on the next run
        st.use_graph = True
        model.sample_from_scratch(*c, **kw)
        model.sample_from_scratch(*c, **kw)
        assert any(e["graph"] is not None for e in st._steps.values())
        with torch.no_grad():
            model.transformer.content_head[1].weight.mul_(-1.0)
        flipped = [r.cpu().numpy() for r in model.sample_from_scratch(*c, **kw)]
        st2 = next(iter(model._decode_states.values()))
        st2.use_graph = False
        st2._steps.clear()
        flipped_eager = [r.cpu().numpy() for r in model.sample_from_scratch(*c, **kw)]
        assert all(np.array_equal(a, b) for a, b in zip(flipped, flipped_eager))
        assert not all(np.array_equal(a, b) for a, b in zip(first, flipped)), "negated content head must change the draws"


def test_sampler_constraints_and_filters_golden(dev):
    from dynamicvectorquantization_amd import stage2
    g = load_golden("sampler")
    model = _sampler_model(dev, "region-first")
    lg = torch.from_numpy(g["h_logits"]).to(dev)
    flag = torch.from_numpy(g["h_flag"]).to(dev)
    sp = torch.from_numpy(g["h_sampled"]).to(dev)
    assert np.array_equal(model.avoid_repeat_or_enforce_pad_for_coarse_position(lg, sp, flag).cpu().numpy(), g["h_coarse"])
    spf = torch.tensor([[66, 3, 7], [66, 0, 65], [66, 5, 5], [66, 14, 2]], device=dev)
    assert np.array_equal(model.avoid_repeat_or_enforce_pad_for_fine_position(lg, spf, flag).cpu().numpy(), g["h_fine"])
    lgc = torch.from_numpy(g["h_logits_c"]).to(dev)
    assert np.array_equal(model.avoid_special_or_enforce_pad_for_content(lgc, flag).cpu().numpy(), g["h_content"])
    assert np.array_equal(stage2.top_k_logits(lgc, 20).cpu().numpy(), g["h_topk"])
    np.testing.assert_allclose(stage2.top_p_logits(torch.softmax(lgc, -1), 0.6).cpu().numpy(), g["h_topp"], rtol=1e-5, atol=1e-8)


def test_class_conditional_dualformer(dev):
    """dqtransformer_class2_entropy variant: class-label start tokens, a train step, and the sampler never emits an id above
    <eos> (content) / above <eos> (fine positions)"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    cfg = dualformer_config()
    p = cfg["params"]
    p["transformer_config"]["params"].update(vocab_size=524, coarse_position_size=28, fine_position_size=76, embd_pdrop=0.0,
                                             resid_pdrop=0.0, attn_pdrop=0.0)
    p["class_cond_stage_config"] = {"target": "modules.dynamic_modules.label_provider.ClassAwareSOSProvider", "params": dict(
        n_classes=10, threshold_content=514, threshold_coarse_position=18, threshold_fine_position=66, coarse_seg_sos=0, fine_seg_sos=1)}
    del p["uncond_stage_config"]
    cfg["target"] = "models.stage2_dynamic.dqtransformer_class2_entropy.Dualformer"
    with rt.compute_dtype_ctx(torch.bfloat16):
        torch.manual_seed(1)
        model = instantiate_from_config(cfg).to(dev)
        model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 0.0, 100, 10
        model.train()
        x = torch.from_numpy(synth.half_flat_images(3, 64, seed=5)).to(dev)
        lab = torch.tensor([1, 4, 9], device=dev)
        tr = Trainer(model, max_steps=2)
        l0 = tr.train_step({"image": x, "class_label": lab}, 0)
        assert torch.isfinite(l0[0])
        model.eval()
        c = model.encode_to_c(lab)
        assert c[0].view(-1).tolist() == [515, 518, 523] and c[2].view(-1).tolist() == [19, 22, 27]
        for fix in (False, True):
            cc, fc, cp, fp = model.sample_from_scratch(*c, sample=True, top_k=20, top_k_pos=10, process=False, fix_fine_position=fix)
            assert int(cc.max()) <= 513 and int(fc.max()) <= 513 and int(fp.max()) <= 65 and int(cp.max()) <= 17
            img = model.decode_to_img(cc, fc, cp, fp)
            assert bool(torch.isfinite(img).all())


def test_sample_many_lanes_equal_sequential(dev):
    """Dualformer.sample_many: five class-conditional batches (different labels, so different sequences) on two, three and four concurrent
    lanes -- own stream, K/V caches, captured token-step graphs per lane -- draw, greedily, exactly the tokens of the sequential
    sampler, batch by batch; multinomial draws stay inside the constraints"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    cfg = dualformer_config()
    p = cfg["params"]
    p["transformer_config"]["params"].update(vocab_size=524, coarse_position_size=28, fine_position_size=76, embd_pdrop=0.0,
                                             resid_pdrop=0.0, attn_pdrop=0.0)
    p["class_cond_stage_config"] = {"target": "modules.dynamic_modules.label_provider.ClassAwareSOSProvider", "params": dict(
        n_classes=10, threshold_content=514, threshold_coarse_position=18, threshold_fine_position=66, coarse_seg_sos=0, fine_seg_sos=1)}
    del p["uncond_stage_config"]
    cfg["target"] = "models.stage2_dynamic.dqtransformer_class2_entropy.Dualformer"
    with rt.compute_dtype_ctx(torch.bfloat16):
        torch.manual_seed(3)
        model = instantiate_from_config(cfg).to(dev).eval()
        with torch.no_grad():
            for n_, p_ in model.transformer.named_parameters():
                if n_.endswith("head.1.weight"):
                    p_.mul_(6.0)                     # spread logits: greedy picks far from ties
        labels = [torch.tensor(l, device=dev) for l in ([1, 4, 9], [0, 2, 3], [5, 5, 8], [7, 6, 1], [9, 0, 4])]
        conds = [model.encode_to_c(l) for l in labels]
        for fix in (True, False):
            kw = dict(sample=False, top_k=20, top_k_pos=10, process=False, fix_fine_position=fix)
            seq = [[t.cpu() for t in model.sample_from_scratch(*c, **kw)] for c in conds]
            assert not all(torch.equal(a, b) for a, b in zip(seq[0], seq[1])), "the batches must differ for this test to mean anything"
            for lanes in (2, 3, 4):       # 4 = the sampling scripts' default
                got = model.sample_many(conds, n_streams=lanes, **kw)
                for i, (a, b) in enumerate(zip(seq, got)):
                    assert all(torch.equal(x_, y_.cpu()) for x_, y_ in zip(a, b)), (fix, lanes, i)
                again = model.sample_many(conds, n_streams=lanes, **kw)        # all lanes warm: every batch runs concurrently
                for i, (a, b) in enumerate(zip(seq, again)):
                    assert all(torch.equal(x_, y_.cpu()) for x_, y_ in zip(a, b)), (fix, lanes, i, "second call")
        outs = model.sample_many(conds, n_streams=2, sample=True, top_k=20, top_k_pos=10, process=False, fix_fine_position=False)
        for cc, fc, cp, fp in outs:
            assert int(cc.max()) <= 513 and int(fc.max()) <= 513 and int(fp.max()) <= 65 and int(cp.max()) <= 17
        torch.cuda.synchronize()
        # a fixed seed reproduces the multinomial samples run to run: batch i always runs on lane i % n_streams, in order, so the
        # per-lane generator streams see the same batches whatever the thread timing (restart the streams = a fresh process)
        draws = []
        for _ in range(3):
            model.__dict__.pop("_sampler_lane_states", None)
            o = model.sample_many(conds, n_streams=2, sample=True, top_k=20, top_k_pos=10, process=False, fix_fine_position=False)
            draws.append([[t.cpu() for t in b] for b in o])
        for other in draws[1:]:
            for a, b in zip(draws[0], other):
                assert all(torch.equal(x_, y_) for x_, y_ in zip(a, b))


def test_sampling_script_end_to_end(dev, tmp_path):
    """scripts/sample_val/sample_dynamic_uncond.py (the reference's flags and output layout) on the shipped p6c18 YAML with random
    weights: 3 samples in batches of 2, fixed fine positions -> two pickles of [B,3,256,256] images in [0,1] and a PNG grid"""
    import os
    import pickle
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "scripts/sample_val/sample_dynamic_uncond.py"), "--yaml_path",
                        "configs/stage2/uncond_imagenet_p6c18.yml", "--batch_size", "2", "--sample_num", "3", "--top_k", "300",
                        "--top_k_pos", "100", "--sample_with_fixed_pos", "--save_image", "--seed", "3", "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    tag = "fixed_TopK-300-100_TopP-1.0-1.0_Temp-1.0"
    pk = sorted(os.listdir(os.path.join(str(tmp_path), tag + "_pickle")))
    assert pk == ["samples_(0_2).pkl", "samples_(1_2).pkl"]
    shapes = []
    for f in pk:
        with open(os.path.join(str(tmp_path), tag + "_pickle", f), "rb") as fp:
            a = pickle.load(fp)
        assert a.dtype == np.float32 and a.min() >= 0.0 and a.max() <= 1.0 and np.isfinite(a).all()
        shapes.append(a.shape)
    assert shapes == [(2, 3, 256, 256), (1, 3, 256, 256)]
    assert sorted(os.listdir(os.path.join(str(tmp_path), tag + "_image"))) == ["batch_0.png", "batch_1.png"]
    assert "token-steps/s" in r.stdout


def test_sample_images_script_end_to_end(dev, tmp_path):
    """scripts/sample_images/sample_dynamic_uncond.py (the reference's PNG-writing twin): one normalised PNG per sample, no pickles"""
    import os
    import subprocess
    import sys
    from PIL import Image
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "scripts/sample_images/sample_dynamic_uncond.py"), "--yaml_path",
                        "configs/stage2/uncond_imagenet_p6c18.yml", "--batch_size", "2", "--sample_num", "3", "--top_k", "300",
                        "--top_k_pos", "100", "--seed", "3", "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    tag = "TopK-300-100_TopP-1.0-1.0_Temp-1.0"
    files = sorted(os.listdir(os.path.join(str(tmp_path), tag + "_image")))
    assert files == ["batch_0_0.png", "batch_0_1.png", "batch_1_0.png"]
    assert not os.path.exists(os.path.join(str(tmp_path), tag + "_pickle"))
    im = np.asarray(Image.open(os.path.join(str(tmp_path), tag + "_image", files[0])))
    assert im.shape == (256, 256, 3) and im.min() == 0 and im.max() == 255          # min-max normalised


def _torch_draw_probs(df, logits, temperature, k, p, rule):
    """filtered distribution of the op-by-op path (the reference's arithmetic) for `rule`"""
    from dynamicvectorquantization_amd.stage2 import top_k_logits, top_p_logits
    kind, sampled, done = rule
    fn = {"coarse_pos": lambda lg: df.avoid_repeat_or_enforce_pad_for_coarse_position(lg, sampled, done),
          "fine_pos": lambda lg: df.avoid_repeat_or_enforce_pad_for_fine_position(lg, sampled, done),
          "content": lambda lg: df.avoid_special_or_enforce_pad_for_content(lg, done)}[kind]
    lg = fn(logits / temperature)
    if k is not None:
        lg = top_k_logits(lg, k)
    pr = torch.softmax(lg, dim=-1)
    if p is not None:
        pr = top_p_logits(pr, p)
    return pr


def test_fused_constrained_sampler_vs_op_by_op(dev):
    """dvq_sample_constrained (mask rules + top-k / top-p + softmax + draw in one launch) against the op-by-op path that the sampler
    goldens pin on the reference: (a) greedy picks are identical for the three rules incl. finished rows; (b) 4000 multinomial draws
    never leave the filtered support and follow its probabilities (total-variation distance); (c) the generator state advances"""
    import types
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.stage2 import _SamplerMixinBase
    df = _SamplerMixinBase.__new__(_SamplerMixinBase)
    df.__dict__.update(coarse_position_pad_code=256, coarse_position_eos_code=257, max_coarse_postion_idx=255, fine_position_pad_code=1024,
                       fine_position_eos_code=1025, fine_position_sos_code=1026, content_pad_code=1024, content_eos_code=1025,
                       content_sos_code=1026)
    g = torch.Generator(device="cpu").manual_seed(5)
    b = 16
    done = torch.zeros(b, 1, device=dev)
    done[3] = 1
    done[11] = 2
    cases = [("coarse_pos", 259, torch.randint(0, 259, (b, 40), generator=g).to(dev)),
             ("fine_pos", 1027, torch.randint(0, 1027, (b, 300), generator=g).to(dev)),
             ("content", 1027, None)]
    for kind, v, sampled in cases:
        logits = (torch.randn(b, v, generator=g) * 3).to(dev)
        rule = (kind, sampled, done)
        for (k, p) in [(None, None), (50, None), (None, 0.9), (100, 0.85)]:
            ref = _torch_draw_probs(df, logits, 0.8, k, p, rule)
            got = df._draw_rule(logits, 0.8, False, k, p, rule).view(-1)
            assert torch.equal(got, ref.argmax(dim=-1)), (kind, k, p)
        # multinomial: support and frequencies (row 0: live; row 3: finished -> always the pad code)
        ref = _torch_draw_probs(df, logits, 1.0, 20, 0.95, rule)
        n_draws = 4000
        counts = torch.zeros(b, v, device=dev)
        s0 = None
        for _ in range(n_draws):
            ix = df._draw_rule(logits, 1.0, True, 20, 0.95, rule)
            counts.scatter_add_(1, ix, torch.ones(b, 1, device=dev))
            s0 = s0 if s0 is not None else int(df._sampler_state[1])
        assert int(df._sampler_state[1]) == s0 + n_draws - 1            # one counter tick per draw
        assert float(counts[ref == 0].sum()) == 0.0, kind               # never outside the filtered support
        tv = 0.5 * (counts / n_draws - ref).abs().sum(dim=1)
        assert float(tv.max()) < 0.06, (kind, tv)                       # 20-entry support, 4000 draws: sampling noise ~0.03
        pad = {"coarse_pos": 256, "fine_pos": 1024, "content": 1024}[kind]
        assert float(counts[3, pad]) == n_draws and float(counts[11, pad]) == n_draws
    # bf16 logits and a padded (strided) logits matrix
    lg = (torch.randn(b, 1032, generator=g) * 2).to(dev).to(torch.bfloat16)
    rule = ("content", None, done)
    got = df._draw_rule(lg[:, :1027], 1.0, False, 10, None, rule).view(-1)
    ref = _torch_draw_probs(df, lg[:, :1027].float(), 1.0, 10, None, rule).argmax(dim=-1)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("mode", ["phases", "persistent"])
@pytest.mark.parametrize("b,wgs,wattn", [(5, 0, 0), (16, 0, 0), (50, 0, 0), (50, 32, 1), (37, 30, 2), (64, 64, 1), (50, 128, 2), (7, 16, 2)])
def test_decode_stack_kernel_vs_per_kernel_token_steps(dev, monkeypatch, b, wgs, wattn, mode):
    """dvq_decode_stack (all blocks of a transformer per token step: five fused launches per block -- `phases`, the default -- or ONE
    persistent kernel with device-wide barriers -- `persistent`) against the per-kernel token steps it replaces: same logits within bf16 rounding over a run of single-row steps, same K/V cache rows, eager and
    replayed from the captured graph; barrier error word stays clear.  b = 50 is the reference's sampling batch
    (scripts/sample_val/sample_dynamic_uncond.py:29): row tiles of 16, and -- with few workgroups (`wgs`) -- the one-wave-per-item
    attention paths (one wave / a pair of waves splitting the cache rows per item); b = 37: a ragged last tile, three tiles."""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.stackgpt import DecodeState
    from dynamicvectorquantization_amd import synth
    cfg = dict(SAMPLER_GPT_CFG, n_embd=256, n_head=4, position_layer=2, content_layer=3)
    steps = 14
    monkeypatch.setenv("DVQ_DECODE_MODE", mode)
    monkeypatch.setenv("DVQ_DECODE_ATTN_SPLIT", "1" if b in (5, 7) else "0")      # two workgroups per attention item + merge in the projection
    monkeypatch.setenv("DVQ_DECODE_WGS", str(wgs))
    monkeypatch.setenv("DVQ_DECODE_WAVE_ATTN", str(wattn))      # attention by the whole workgroup / one wave / a pair of waves per item
    with rt.compute_dtype_ctx(torch.bfloat16):
        gpt = instantiate_from_config({"target": "modules.dynamic_modules.stackgpt.StackGPT", "params": cfg}).to(dev).eval()
        with torch.no_grad():
            for n, p in gpt.named_parameters():
                v = synth.det_param("decstack." + n, tuple(p.shape))
                p.copy_(torch.from_numpy(v * (0.3 if n == "pos_emb" else 1.0)).to(dev))
        rs = np.random.RandomState(5)
        tok_c = torch.from_numpy(rs.randint(0, 512, (b, steps))).to(dev)
        tok_p = torch.from_numpy(rs.randint(0, 16, (b, steps))).to(dev)
        tok_u = torch.from_numpy(rs.randint(0, 16, (b, steps))).to(dev)
        seg = torch.zeros(b, steps, dtype=torch.long, device=dev)
        table = gpt.content_coarse_pos_emb.weight

        def run(flag, use_graph):
            monkeypatch.setenv("DVQ_DECODE_STACK", flag)
            st = DecodeState(gpt, b, 64)
            st.use_graph = use_graph
            outs = []
            for i in range(steps):
                pl = st.position_rows(tok_c[:, i:i + 1], tok_p[:, i:i + 1], table, None, seg[:, i:i + 1] if gpt.activate_segment else None)
                cl = st.content_rows(tok_u[:, i:i + 1], table)
                outs.append((pl.float().cpu().numpy(), cl.float().cpu().numpy()))
            if flag == "1":
                assert st._stacks, "the fused kernel was not used"
                st.check()                                       # raises if a device-wide barrier timed out
            caches = [c_[0][:, :steps].float().cpu().numpy() for c_ in st.con_cache] + [c_[1][:, :steps].float().cpu().numpy() for c_ in st.pos_cache]
            return outs, caches

        ref, ref_caches = run("0", False)
        for use_graph in (False, True):
            got, caches = run("1", use_graph)
            for i, ((pr, cr), (pg, cg)) in enumerate(zip(ref, got)):
                for name, a, g_ in (("position", pr, pg), ("content", cr, cg)):
                    err = float(np.abs(a - g_).max()) / max(1e-6, float(np.abs(a).max()))
                    assert err < 4e-2, f"step {i} {name} logits: rel-to-max error {err} (graph={use_graph})"
            for a, g_ in zip(ref_caches, caches):
                err = float(np.abs(a - g_).max()) / max(1e-6, float(np.abs(a).max()))
                assert err < 4e-2, f"cache rows differ: {err}"
