"""GPU parity of the COMPLETE training step against the reference (tests/golden/train_step_*.npz, captured by driving the real
reference through Lightning's two-optimizer order: tools/gen_golden.py::run_reference_train_steps) and of the fused Adam / AdamW
kernels against torch.optim.  `pytest -m gpu`.

What is compared per step (same keys and the same distance functions as the oracle's pin, oracle/train_step.py::compare_records):
code indices of BOTH autoencoder forwards, cluster_size_ema / embed_ema / codebook after each of the TWO EMA updates per batch, every
loss term the reference logs, the learning rate, gradients + Adam moments + parameter movement of 20 watched tensors (autoencoder and
discriminator), PatchGAN BatchNorm running statistics.  dqvae_dual_entropy.py:154-183,206-231; quantize2_mask.py:66-126."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from dynamicvectorquantization_amd import synth
from golden_cfg import TRAIN_STEP, TRAIN_STEP_WATCH, train_step_stride

pytestmark = pytest.mark.gpu


def _sample(t):
    a = t.detach().reshape(-1).float().cpu().numpy()
    return a[:: train_step_stride(a.size)].astype(np.float32).copy()


def run_hip_train_steps(tag, dev, mode, use_graph=False):
    """the pinned run on the HIP trainer -> {fixture key: value}"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    from test_gpu_model import GEOM, model_config
    triple = tag == "triple"
    if triple:
        from golden_cfg import TRAIN_STEP_TRIPLE, TRAIN_STEP_TRIPLE_WATCH, train_step_triple_lossconfig
        from test_gpu_featrouted import feat_model_config
        c = TRAIN_STEP_TRIPLE
        g = dict(k=c["k"], zc=c["zc"], resolution=64)
        watch = TRAIN_STEP_TRIPLE_WATCH
        gold = load_golden("train_step_triple")
    else:
        c = TRAIN_STEP[tag]
        g = GEOM[c["geom"]]
        watch = TRAIN_STEP_WATCH
    k, zc = g["k"], g["zc"]
    out = {}
    with rt.compute_dtype_ctx(mode):
        torch.manual_seed(0)
        if triple:
            model = instantiate_from_config(feat_model_config("triple", k=k, zc=zc, loss=train_step_triple_lossconfig(c["ndf"]))).to(dev)
            synth.apply_train_step_state(model, k, zc, scale={"encoder.router.gate.2.weight": c["last_gate_scale"]})
        else:
            model = instantiate_from_config(model_config(**g, loss="full", ndf=c["ndf"])).to(dev)
            synth.apply_train_step_state(model, k, zc)
        rt.bump_weights_epoch()
        model.learning_rate, model.min_learning_rate = c["lr"], c["min_lr"]
        model.warmup_epochs, model.steps_per_epoch, model.training_steps = c["warmup_epochs"], c["steps_per_epoch"], c["training_steps"]
        model.train()
        tr = Trainer(model, max_steps=c["steps"], use_graph=use_graph)
        assert len(tr.opts) == 2
        params = dict(model.named_parameters())
        cbm = model.quantize.codebook

        def snap(pre, loss=None):
            out[pre + "codes"] = model._last["codes"].reshape(-1).cpu().numpy().astype(np.int16)
            out[pre + "cluster_size_ema"] = cbm.cluster_size_ema.cpu().numpy().copy()
            out[pre + "embed_ema"] = cbm.embed_ema.cpu().numpy()[:: k // 64].copy()
            out[pre + "codebook"] = cbm.weight.detach().cpu().numpy()[:-1][:: k // 64].copy()

        cur = {"step": 0}
        orig_steps = [o.step for o in tr.opts]

        def wrap(oi):
            def step_and_record(closure=None):
                s = cur["step"]
                pre = f"s{s}.o{oi}."
                out[pre + "lr"] = np.float64(tr.opts[oi].param_groups[0]["lr"])
                snap(pre)                     # the EMA update of this optimizer's forward has run; the next forward has not
                if s == 0:
                    for n_ in watch:
                        if n_.startswith("loss.discriminator.") == (oi == 1):
                            out[f"s0.grad.{n_}"] = _sample(params[n_].grad)
                return orig_steps[oi](closure)
            return step_and_record

        for oi, o in enumerate(tr.opts):
            o.step = wrap(oi)
        for step, xb in enumerate(synth.train_step_batches(c["steps"], c["bs"], g["resolution"])):
            cur["step"] = step
            if triple:              # the fixture's inputs: restart permutation (from the reference's grain map) and the Gumbel noise
                cbm.restart_perm = torch.from_numpy(gold[f"s{step}.perm"].astype(np.int64))
                model.encoder.gumbel_exponential = torch.from_numpy(synth.train_step_gumbel(step, c["bs"])).to(dev)
            else:
                cbm.restart_perm = torch.from_numpy(synth.train_step_restart_perm(step, c["bs"], k, g["resolution"]))
            losses = tr.train_step({"image": torch.from_numpy(xb).to(dev)}, step)
            torch.cuda.synchronize()
            for oi, l in enumerate(losses):
                out[f"s{step}.o{oi}.loss"] = np.float32(float(l))
            for k_, v_ in model._logged.items():
                out[f"s{step}.log.{k_}"] = np.float32(float(v_))
            states = [tr._optimizer_state_dict(o) for o in tr.opts]
            index = [{id(p): i for i, p in enumerate(p_ for grp in o.param_groups for p_ in grp["params"])} for o in tr.opts]
            for n_ in watch:
                oi = 1 if n_.startswith("loss.discriminator.") else 0
                st = states[oi]["state"][index[oi][id(params[n_])]]
                assert int(float(st["step"])) == step + 1
                out[f"s{step}.param.{n_}"] = _sample(params[n_])
                out[f"s{step}.exp_avg.{n_}"] = _sample(st["exp_avg"])
                out[f"s{step}.exp_avg_sq.{n_}"] = _sample(st["exp_avg_sq"])
        for n_, b in model.loss.discriminator.named_buffers():
            out["final.disc_buf." + n_] = b.detach().cpu().numpy().copy()
    return out


# worst allowed distance per group (oracle/train_step.py::summarize), from measurements on MI355X (gpurun_out/test_reports.jsonl, kept in
# profiles/r06_train_step_parity.txt); the oracle's own distance to the reference (both fp32 on the host) is oracle.train_step.PIN_BOUNDS.
# Steps 0 and 1 start from bit-identical parameters (step 0 runs at lr 0), so they are held to rounding level; step 2 starts from
# parameters that already carry Adam's sign-level differences (`dparam` of step 1) and may flip a few codes whose fp64 top-2 gap is < 1e-4.
def _bounds(mode):
    from oracle.train_step import PIN_BOUNDS
    b = dict(PIN_BOUNDS)                # fp32 (exact fp32 matrix instructions): the same bounds the host oracle is pinned with
    if mode == "fp32x3":                # products at ~2^-17: rounding-level near ties may flip from the first forward on, and the
        #                                 ill-conditioned discriminator gradients (PIN_BOUNDS) carry ~10 % (measured 0.11 / 0.18)
        b.update({"codes_near": 2, "cluster_size_ema": 1e-3, "scalar:loss": 1e-3, "scalar": 3e-2,
                  "scalar:train_logits_fake": 0.3, "scalar:train_logits_real": 0.3, "s2:scalar:train_logits_fake": 0.3, "s2:scalar:train_logits_real": 0.3,
                  "grad": 0.1, "exp_avg": 0.1, "exp_avg_sq": 0.1, "s2:exp_avg": 0.2, "s2:exp_avg_sq": 0.2,
                  "grad_disc": 0.25, "exp_avg_disc": 0.25, "exp_avg_sq_disc": 0.35, "s2:exp_avg_disc": 0.35, "s2:exp_avg_sq_disc": 0.35,
                  "dparam_disc": 0.35})
    return b


@pytest.mark.parametrize("mode", ["fp32", "fp32x3"])
@pytest.mark.parametrize("tag", ["small", "c1"])
def test_train_step_golden(dev, tag, mode):
    from oracle import train_step as ots
    from test_gpu_model import GEOM, _report
    g = load_golden(f"train_step_{tag}")
    ref = {k_: g[k_] for k_ in g.files}
    meta = {k_: g[k_] for k_ in ("state_keys", "state_shapes", "param_keys")}
    geom = GEOM[TRAIN_STEP[tag]["geom"]]
    got = run_hip_train_steps(tag, dev, mode)
    missing = [k_ for k_ in ref if k_ not in got and not k_.startswith(("state_", "param_keys")) and not k_.endswith(".gap")]
    assert not missing, missing
    cmp = ots.compare_records(got, ref, start_param=ots.sampled_start_param(meta, geom["k"], geom["zc"], train_step_stride))
    summ = ots.summarize(cmp)
    _report("train_step_golden", tag=tag, mode=mode, **{f"{s}.{grp}": float(e) for (s, grp), e in sorted(summ.items())})
    _report("train_step_golden_keys", tag=tag, mode=mode, **{k_: float(e) for k_, (_, e) in sorted(cmp.items()) if ".log." not in k_})
    bad = ots.check_summary(summ, _bounds(mode))
    assert not bad, bad


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_train_step_bf16_distance_report(dev, tag):
    """the BENCHMARKED precision against the same fixtures: bf16 cannot meet the fp32 bounds (north_star's tolerance is missed by the bf16
    forward already), so this test REPORTS its distances (gpurun_out/test_reports.jsonl -> profiles/) and asserts only
    what must hold at any precision: the lr = 0 step leaves every parameter untouched, the learning rates are the reference's, losses are
    within 5 % and the Adam moments' relative L2 distance stays below 1"""
    from oracle import train_step as ots
    from test_gpu_model import GEOM, _report
    g = load_golden(f"train_step_{tag}")
    ref = {k_: g[k_] for k_ in g.files}
    meta = {k_: g[k_] for k_ in ("state_keys", "state_shapes", "param_keys")}
    geom = GEOM[TRAIN_STEP[tag]["geom"]]
    got = run_hip_train_steps(tag, dev, torch.bfloat16)
    summ = ots.summarize(ots.compare_records(got, ref, start_param=ots.sampled_start_param(meta, geom["k"], geom["zc"], train_step_stride)))
    _report("train_step_bf16_distance", tag=tag, **{f"{s}.{grp}": float(e) for (s, grp), e in sorted(summ.items())})
    for (s, grp), e in summ.items():
        if grp.startswith("dparam0") or grp == "scalar:lr":
            assert e <= 1e-12, (s, grp, e)
        if grp == "scalar:loss":
            assert e < 5e-2, (s, grp, e)
        if grp in ("exp_avg", "grad"):
            # (the shrunken model's watched gradients are dominated by the noise-sized GAN / BatchNorm residual: 0.89 - 0.95 from run to
            #  run -- fp32 atomics arrive in a different order --, once past 1.0; the full-width model sits at 0.79 - 0.80)
            assert e < (1.5 if tag == "small" else 1.0), (s, grp, e)


@pytest.mark.parametrize("mode", ["fp32", "fp32x3"])
def test_triple_train_step_golden(dev, mode):
    """the feature-routed TRIPLE-grain model (BASELINE config 4's family) through its two-optimizer step against the reference
    (tests/golden/train_step_triple.npz): Gumbel straight-through routing with the injected noise, budget loss on the gate, router
    parameters trained by optimizer 0; dqvae_triple_feat.py:102-136,164-196"""
    from golden_cfg import TRAIN_STEP_TRIPLE as C
    from oracle import train_step as ots
    from test_gpu_model import _report
    g = load_golden("train_step_triple")
    ref = {k_: g[k_] for k_ in g.files}
    got = run_hip_train_steps("triple", dev, mode)
    skip = tuple(k_ for k_ in ref if k_.endswith((".perm", ".grain")))
    missing = [k_ for k_ in ref if k_ not in got and not k_.startswith(("state_", "param_keys")) and not k_.endswith(".gap") and k_ not in skip]
    assert not missing, missing
    p0 = ots.sampled_start_param(ref, C["k"], C["zc"], train_step_stride, scale={"encoder.router.gate.2.weight": C["last_gate_scale"]})
    summ = ots.summarize(ots.compare_records(got, ref, start_param=p0, skip=skip))
    _report("triple_train_step_golden", mode=mode, **{f"{s}.{grp}": float(e) for (s, grp), e in sorted(summ.items())})
    bad = ots.check_summary(summ, _bounds(mode))
    assert not bad, bad


def test_train_step_golden_graph_replay_matches_eager(dev):
    """the recorded (hipGraph) training step must walk the same trajectory as the eager one the golden test pins: same run without
    the injected permutation (a host-side injection keeps the step eager), eager vs recorded, parameters and EMA state after 6 steps"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    from test_gpu_model import GEOM, model_config
    c, g = TRAIN_STEP["small"], GEOM["small"]
    res = []
    for use_graph in (False, True):
        with rt.compute_dtype_ctx("fp32"):
            torch.manual_seed(0)
            model = instantiate_from_config(model_config(**g, loss="full", ndf=c["ndf"])).to(dev)
            synth.apply_train_step_state(model, g["k"], g["zc"])
            rt.bump_weights_epoch()
            model.learning_rate, model.min_learning_rate = c["lr"], c["min_lr"]
            model.warmup_epochs, model.steps_per_epoch, model.training_steps = 0.3, 10, 50
            model.train()
            tr = Trainer(model, max_steps=6, use_graph=use_graph, graph_after=2)
            xs = [torch.from_numpy(xb).to(dev) for xb in synth.train_step_batches(6, c["bs"], g["resolution"])]
            ls = [[float(l) for l in tr.train_step({"image": x}, i)] for i, x in enumerate(xs)]
            torch.cuda.synchronize()
            if use_graph:
                assert tr.graph_replays >= 2
            res.append((ls, model.decoder.conv_out.weight.detach().clone(), model.quantize.codebook.cluster_size_ema.clone(),
                        model.loss.discriminator.main[0].weight.detach().clone()))
    (l0, w0, n0, d0), (l1, w1, n1, d1) = res
    np.testing.assert_allclose(np.array(l0), np.array(l1), rtol=2e-3)
    assert float((w0 - w1).abs().max()) <= 2.5 * c["lr"] * 6
    assert float((n0 - n1).abs().max() / n0.abs().max()) < 5e-2


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_two_stream_loss_schedule_matches_single_stream(dev, mode, monkeypatch):
    """losses.VQLPIPSWithDiscriminator on two streams (PatchGAN branch of the generator loss on the side stream, target-only work
    prefetched beside the autoencoder's forward: LPIPS on B images against stored target taps instead of one 2 B batch) must give
    the step the single-stream schedule gives: every image's values are independent of the batch it is evaluated in.  Compared on
    ONE step from identical parameters (later steps carry Adam's sign-level amplification of rounding noise and code flips, which the
    golden tests bound separately): both losses and every parameter gradient at the moment its optimizer steps."""
    from dynamicvectorquantization_amd import losses as L
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    from test_gpu_model import GEOM, model_config
    c, g = TRAIN_STEP["small"], GEOM["small"]
    res = []
    for two_stream in (True, False):
        monkeypatch.setattr(L, "_GEN_SIDE", two_stream)
        monkeypatch.setattr(L, "_LOSS_PREFETCH", two_stream)
        with rt.compute_dtype_ctx(mode):
            torch.manual_seed(0)
            model = instantiate_from_config(model_config(**g, loss="full", ndf=c["ndf"])).to(dev)
            synth.apply_train_step_state(model, g["k"], g["zc"])
            rt.bump_weights_epoch()
            model.learning_rate, model.min_learning_rate = c["lr"], c["min_lr"]
            model.warmup_epochs, model.steps_per_epoch, model.training_steps = 0.3, 10, 50
            model.train()
            tr = Trainer(model, max_steps=1, use_graph=False)
            grads = {}
            names = {id(p): n for n, p in model.named_parameters()}

            def wrap(o, orig):
                def step(closure=None):
                    for grp in o.param_groups:
                        for p in grp["params"]:
                            if p.grad is not None:
                                grads[names[id(p)]] = p.grad.detach().float().clone()
                    return orig(closure)
                return step

            for o in tr.opts:
                o.step = wrap(o, o.step)
            x = torch.from_numpy(next(iter(synth.train_step_batches(1, c["bs"], g["resolution"])))).to(dev)
            ls = [float(l) for l in tr.train_step({"image": x}, 0)]
            torch.cuda.synchronize()
            res.append((ls, grads))
    (l1, g1), (l0, g0) = res
    np.testing.assert_allclose(np.array(l1), np.array(l0), rtol=1e-5 if mode == "fp32" else 1e-4)
    assert set(g1) == set(g0) and len(g0) > 100
    # rms difference per tensor relative to the tensor's rms gradient; tensors whose exact gradient is zero (conv biases in front of a
    # normalisation: pure rounding noise in both runs) are measured against 1e-3 of the largest rms gradient of the model instead
    rms = {n: float(g0[n].pow(2).mean().sqrt()) for n in g0}
    floor = 1e-3 * max(rms.values())
    worst, worst_noise = ("", 0.0), ("", 0.0)
    for n in g0:
        d = float((g1[n] - g0[n]).pow(2).mean().sqrt())
        if rms[n] >= floor:
            worst = max(worst, (n, d / rms[n]), key=lambda t: t[1])
        else:
            worst_noise = max(worst_noise, (n, d / floor), key=lambda t: t[1])
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    from test_gpu_model import _report
    _report("two_stream_vs_single_stream_gradients", mode=mode, worst=worst, worst_noise=worst_noise, global_rel=(num / den) ** 0.5)
    # fp32: atomic / split-reduction order.  bf16: one-ulp differences of the BatchNorm statistics re-round activations, and fifty
    # layers of backward amplify that on the reduction-heavy tensors (biases of the first layers): bounded loosely per tensor, tightly
    # over all gradient elements
    assert worst[1] <= (1e-4 if mode == "fp32" else 0.3), worst
    assert (num / den) ** 0.5 <= (1e-5 if mode == "fp32" else 3e-2), (num / den) ** 0.5
    assert worst_noise[1] <= (1e-2 if mode == "fp32" else 1.0), worst_noise      # i.e. below 1e-5 (bf16: 1e-3) of the largest rms gradient


# ---- the fused optimizer kernels against torch.optim ----------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["adam", "adamw"])
def test_hip_adam_matches_torch_optim(dev, kind):
    """HipAdam (dvq_adamw_dev on the flat buffers, dvq_adam per tensor) vs torch.optim.Adam / AdamW over 4 steps: stage 1's betas (0.5, 0.9)
    without decay, stage 2's (0.9, 0.95) with decay / no-decay groups; a tensor whose gradient is exactly zero, a 4-D conv weight (stored
    channel-last in the flat buffer), a 1-element tensor (unaligned offsets behind it), a learning rate that changes every step"""
    from dynamicvectorquantization_amd.trainer import HipAdam
    rs = np.random.RandomState(1)
    shapes = [(37, 5), (1,), (8, 4, 3, 3), (11,), (6, 6), (130, 7)]
    p0 = [rs.standard_normal(s).astype(np.float32) * (1e-3 if i == 3 else 1.0) for i, s in enumerate(shapes)]
    betas, wd = ((0.5, 0.9), 0.0) if kind == "adam" else ((0.9, 0.95), 0.01)
    for flat in (True, False):
        ref_p = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in p0]
        hip_p = [torch.nn.Parameter(torch.from_numpy(a.copy()).to(dev)) for a in p0]
        if kind == "adam":
            ref_opt = torch.optim.Adam(ref_p, lr=1e-3, betas=betas)
            hip_opt = HipAdam(hip_p, lr=1e-3, betas=betas)
        else:
            ref_opt = torch.optim.AdamW([{"params": ref_p[:3], "weight_decay": wd}, {"params": ref_p[3:], "weight_decay": 0.0}], lr=1e-3, betas=betas)
            hip_opt = HipAdam([{"params": hip_p[:3], "weight_decay": wd}, {"params": hip_p[3:], "weight_decay": 0.0}], lr=1e-3, betas=betas)
        if flat:
            hip_opt.flatten()
        for t in range(1, 5):
            lr = 1e-3 * (0.5 + 0.25 * t)
            for opt in (ref_opt, hip_opt):
                for grp in opt.param_groups:
                    grp["lr"] = lr
            grads = [rs.standard_normal(s).astype(np.float32) * (10.0 ** rs.randint(-6, 2)) for s in shapes]
            grads[4][:] = 0.0
            for q, h, gq in zip(ref_p, hip_p, grads):
                q.grad = torch.from_numpy(gq.copy())
                if flat:
                    h.grad.copy_(torch.from_numpy(gq).to(dev))
                else:
                    h.grad = torch.from_numpy(gq.copy()).to(dev)
            ref_opt.step()
            hip_opt.step()
            torch.cuda.synchronize()
            for i, (q, h) in enumerate(zip(ref_p, hip_p)):
                np.testing.assert_allclose(h.detach().cpu().numpy(), q.detach().numpy(), rtol=3e-6, atol=1e-9, err_msg=f"{kind} flat={flat} t={t} tensor {i}")
        # moments
        if flat:
            m, v = hip_opt._fstate["m"], hip_opt._fstate["v"]
            off = 0
            from dynamicvectorquantization_amd.trainer import FlatParams
            for q, h in zip(ref_p, hip_p):
                ea = ref_opt.state[q]["exp_avg"].numpy()
                ev = ref_opt.state[q]["exp_avg_sq"].numpy()
                np.testing.assert_allclose(FlatParams._view(m, off, h).cpu().numpy(), ea, rtol=3e-6, atol=1e-6 * max(1e-30, float(np.abs(ea).max())))
                np.testing.assert_allclose(FlatParams._view(v, off, h).cpu().numpy(), ev, rtol=3e-6, atol=1e-30)
                off += h.numel()
        else:
            for q, h in zip(ref_p, hip_p):
                ea = ref_opt.state[q]["exp_avg"].numpy()
                np.testing.assert_allclose(hip_opt.state[h]["exp_avg"].cpu().numpy(), ea, rtol=3e-6, atol=1e-6 * max(1e-30, float(np.abs(ea).max())))
                np.testing.assert_allclose(hip_opt.state[h]["exp_avg_sq"].cpu().numpy(), ref_opt.state[q]["exp_avg_sq"].numpy(), rtol=3e-6, atol=1e-30)


# ---- stage 2: Dualformer + AdamW --------------------------------------------------------------------------------------------------
S2_BOUNDS = {
    # fp32 = exact fp32 matrix instructions; fp32x3 = three bf16 MFMA passes on split operands (~2^-17 per product)
    # measured on MI355X: fp32 4.7e-7 / 7.0e-7 / 2.1e-6 (grad / exp_avg_sq / dparam), fp32x3 1.2e-5 / 1.7e-5 / 4.4e-5
    "fp32": {"scalar:lr": 1e-12, "scalar:loss": 1e-5, "scalar": 1e-5, "grad": 1e-5, "exp_avg": 1e-5, "exp_avg_sq": 2e-5, "dparam0": 0.0, "dparam": 5e-5},
    "fp32x3": {"scalar:lr": 1e-12, "scalar:loss": 1e-5, "scalar": 2e-5, "grad": 1e-4, "exp_avg": 1e-4, "exp_avg_sq": 2e-4, "dparam0": 0.0, "dparam": 5e-4},
}


@pytest.mark.parametrize("mode", ["fp32", "fp32x3"])
def test_dualformer_train_steps_golden(dev, mode):
    """the reference Dualformer's training_step -> AdamW -> LambdaLR for 3 steps (tests/golden/train_step_dualformer.npz;
    dqtransformer_uncond_entropy.py:92-143,217-234) on the HIP trainer: losses, lr, gradients, moments, parameter movement"""
    from conftest import REPO
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    from golden_cfg import TRAIN_STEP_S2, TRAIN_STEP_S2_WATCH, dualformer_cfg, train_step_s2_batch
    from oracle import train_step as ots
    from test_gpu_model import _report
    from test_oracle_golden import dqvae_state_dict
    g = load_golden("train_step_dualformer")
    ref = {k_: g[k_] for k_ in g.files}
    c = TRAIN_STEP_S2
    thr_json = os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json")
    out = {}
    with rt.compute_dtype_ctx(mode):
        cfg = dualformer_cfg("uncond", thr_json)
        cfg["weight_decay"], cfg["warmup_epochs"] = c["weight_decay"], c["warmup_epochs"]
        model = instantiate_from_config({"target": "models.stage2_dynamic.dqtransformer_uncond_entropy.Dualformer", "params": cfg}).to(dev)
        model.first_stage_model.load_state_dict(dqvae_state_dict(load_golden("dqvae_small"), "spread", 512, 64))
        with torch.no_grad():
            for n, p in model.transformer.named_parameters():
                v = synth.det_param("dualformer.uncond." + n, tuple(p.shape))
                p.copy_(torch.from_numpy(v * (0.3 if n == "pos_emb" else 1.0)).to(dev))
        rt.bump_weights_epoch()
        model.learning_rate, model.min_learning_rate = c["lr"], c["min_lr"]
        model.steps_per_epoch, model.training_steps = c["steps_per_epoch"], c["training_steps"]
        model.train()
        tr = Trainer(model, max_steps=c["steps"], use_graph=False)
        (opt,) = tr.opts
        params = dict(model.transformer.named_parameters())
        decay_ids = {id(p) for p in opt.param_groups[0]["params"]}
        assert sorted(n for n, p in params.items() if id(p) in decay_ids) == [str(n) for n in g["decay_names"]]
        orig_step = opt.step
        cur = {"step": 0}

        def step_and_record(closure=None):
            if cur["step"] == 0:
                for n_ in TRAIN_STEP_S2_WATCH:
                    out[f"s0.grad.{n_}"] = _sample(params[n_].grad)
            return orig_step(closure)
        opt.step = step_and_record
        for step in range(c["steps"]):
            cur["step"] = step
            out[f"s{step}.lr"] = np.float64(opt.param_groups[0]["lr"])
            (loss,) = tr.train_step({"image": torch.from_numpy(train_step_s2_batch(step)).to(dev)}, step)
            torch.cuda.synchronize()
            out[f"s{step}.loss"] = np.float32(float(loss))
            for k_, v_ in model._logged.items():
                out[f"s{step}.log.{k_}"] = np.float32(float(v_))
            state = tr._optimizer_state_dict(opt)
            index = {id(p): i for i, p in enumerate(p_ for grp in opt.param_groups for p_ in grp["params"])}
            for n_ in TRAIN_STEP_S2_WATCH:
                st = state["state"][index[id(params[n_])]]
                out[f"s{step}.param.{n_}"] = _sample(params[n_])
                out[f"s{step}.exp_avg.{n_}"] = _sample(st["exp_avg"])
                out[f"s{step}.exp_avg_sq.{n_}"] = _sample(st["exp_avg_sq"])
    missing = [k_ for k_ in ref if k_ not in out and not k_.startswith(("state_", "decay_names"))]
    assert not missing, missing
    summ = ots.summarize(ots.compare_records(out, ref, start_param=ots.dualformer_start_param(ref)))
    _report("dualformer_train_steps_golden", mode=mode, **{f"{s}.{grp}": float(e) for (s, grp), e in sorted(summ.items())})
    bad = ots.check_summary(summ, S2_BOUNDS[mode])
    assert not bad, bad
