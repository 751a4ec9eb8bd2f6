"""Training-step capture (runtime.StepGraph / Trainer): a step replayed from recorded hipGraph segments must do what the
eager step does -- same losses, parameters, Adam state, LR-schedule position -- including with the data-parallel exchange
points (RCCL all-reduces between graph segments), exercised here on ONE GPU through a one-rank nccl group.  `pytest -m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO
from dynamicvectorquantization_amd import synth

pytestmark = pytest.mark.gpu


def _make(dev, use_graph, loss="full", graph_after=2, seed=0):
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    from test_gpu_model import GEOM, model_config
    torch.manual_seed(seed)
    model = instantiate_from_config(model_config(**GEOM["small"], loss=loss)).to(dev)
    # a schedule that moves every step (warm-up then cosine): the replayed optimizer must follow it from device memory
    model.learning_rate, model.min_learning_rate = 2e-4, 1e-5
    model.training_steps, model.steps_per_epoch, model.warmup_epochs = 12, 4, 1
    model.train()
    return model, Trainer(model, max_steps=12, use_graph=use_graph, graph_after=graph_after)


def _run(dev, use_graph, steps, loss="full"):
    from dynamicvectorquantization_amd import runtime as rt
    xs = [torch.from_numpy(synth.half_flat_images(2, 64, seed=40 + i)).to(dev) for i in range(3)]
    with rt.compute_dtype_ctx(torch.float32):
        model, tr = _make(dev, use_graph, loss)
        losses = []
        for i in range(steps):
            out = tr.train_step({"image": xs[i % 3]}, i)
            losses.append([float(l) for l in out])
        torch.cuda.synchronize()
    return model, tr, np.array(losses)


@pytest.mark.parametrize("loss", ["ae", "full"])
def test_step_graph_matches_eager(dev, loss):
    steps = 8
    m_e, tr_e, l_e = _run(dev, False, steps, loss)
    m_g, tr_g, l_g = _run(dev, True, steps, loss)
    assert tr_e.graph_replays == 0
    assert tr_g._graph is not None and tr_g.graph_replays == steps - 2, (tr_g.use_graph, tr_g.graph_replays)
    assert tr_g._graph["sg"].n_segments() == 1            # one rank: the whole two-optimizer step is ONE graph
    # host-side bookkeeping the replay has to carry by hand
    assert m_g.global_step == m_e.global_step == steps
    for og, oe in zip(tr_g.opts, tr_e.opts):
        assert og._fstate["step"] == oe._fstate["step"] == steps
        assert [g["lr"] for g in og.param_groups] == [g["lr"] for g in oe.param_groups]
    # same trajectory.  Autoencoder-only objective: tight (fp32 kernels; the order of atomics is the only difference between two
    # runs).  Complete objective: Adam turns the sign noise of near-zero gradients into +-lr steps and the adaptive GAN weight is
    # a ratio of two gradient norms, so two EAGER runs from one seed already differ by ~1 % after a few steps
    # (tools/debug/graph_aa.py prints that A/A spread next to graph-vs-eager): bounded loosely here, exactly in the `ae` case.
    tl, tp = (2e-3, 2e-3) if loss == "ae" else (1.5e-1, 5e-2)
    np.testing.assert_allclose(l_g[:2], l_e[:2], rtol=1e-6)                # the eager steps before the recording are the same code
    np.testing.assert_allclose(l_g, l_e, rtol=tl, atol=tl / 10)
    if loss == "ae":
        # parameters: relative bound + the slack Adam's sign noise can produce on (near) zero-gradient entries: a few % of the
        # lr * steps each of them could have moved at all.  (Complete objective: trajectories of two runs drift apart -- dead-code
        # restarts pick other rows, a zero-initialised bias differs by 100 % -- only the losses are compared.)
        lr_max = 2e-4
        for (n1, p1), (_, p2) in zip(m_g.named_parameters(), m_e.named_parameters()):
            a, b = p1.detach().float(), p2.detach().float()
            assert float((a - b).norm()) <= tp * float(b.norm()) + 0.05 * lr_max * steps * a.numel() ** 0.5, n1
        for og, oe in zip(tr_g.opts, tr_e.opts):
            assert float((og._fstate["m"] - oe._fstate["m"]).norm()) <= 10 * tp * float(oe._fstate["m"].norm()) + 1e-9
        for k in ("quantize.codebook.cluster_size_ema", "quantize.codebook.embed_ema"):
            a, b = m_g.state_dict()[k].float(), m_e.state_dict()[k].float()
            assert float((a - b).norm()) <= 5e-2 * float(b.norm()) + 1e-6, k
    if loss == "full":     # BatchNorm running statistics / batch counters of the discriminator advance inside the graph
        nb_g = m_g.state_dict()["loss.discriminator.main.3.num_batches_tracked"]
        nb_e = m_e.state_dict()["loss.discriminator.main.3.num_batches_tracked"]
        assert int(nb_g) == int(nb_e) > 0
    # logged scalars are refreshed by the replays
    assert abs(float(m_g._logged["train_fine_ratio"]) - float(m_e._logged["train_fine_ratio"])) < 1e-6


def test_step_graph_signature_change_falls_back(dev):
    """a different batch shape (or an eval/profile step) must not replay the recorded step"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    with rt.compute_dtype_ctx(torch.bfloat16):
        model, tr = _make(dev, True, "ae")
        x2 = torch.from_numpy(synth.half_flat_images(2, 64, seed=1)).to(dev)
        x4 = torch.from_numpy(synth.half_flat_images(4, 64, seed=2)).to(dev)
        for i in range(4):
            tr.train_step({"image": x2}, i)
        assert tr._graph is not None and tr.graph_replays == 2
        out = tr.train_step({"image": x4}, 4)                   # other batch size: eager, recorded graph dropped
        assert tr._graph is None and tr.graph_replays == 2 and all(bool(torch.isfinite(l)) for l in out)
        for i in range(5, 9):
            tr.train_step({"image": x4}, i)
        assert tr._graph is not None and tr.graph_replays >= 4
        K.profile_count_start()                                 # a profiled step runs eagerly (per-launch HIP events)
        r0 = tr.graph_replays
        tr.train_step({"image": x4}, 9)
        assert K.profile_count_stop() > 50 and tr.graph_replays == r0
        torch.cuda.synchronize()


def test_sample_rows_is_a_permutation_prefix(dev):
    from dynamicvectorquantization_amd import kernels as K
    st = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
    for n, k in ((65536, 1024), (1000, 1000), (5, 3), (70000, 8192)):
        a = K.sample_rows(k, n, st).cpu().numpy()
        assert a.min() >= 0 and a.max() < n and len(np.unique(a)) == k
        b = K.sample_rows(k, n, st).cpu().numpy()               # the device counter advanced: another draw
        assert len(np.unique(b)) == k and (n < 10 or not np.array_equal(a, b))
    assert int(st[1]) == 8
    # roughly uniform over [0, n)
    a = K.sample_rows(8192, 65536, st).cpu().numpy()
    assert abs(a.mean() / 65536 - 0.5) < 0.02


def test_step_graph_with_exchange_breaks_one_rank_nccl(dev):
    """the data-parallel step (RCCL all-reduces launched between graph segments, one of them from inside the backward) on
    a ONE-rank nccl group: same results as the eager data-parallel step, every bucket reduced exactly once per step"""
    env = dict(os.environ, DVQ_FORCE_DP="1", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 1000)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp_graph_check.py"), str(port), "both"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_eager_forwards_after_replays_see_current_weights(dev):
    """a replayed step rewrites codebook and parameters without running Python, so the search-side codebook
    planes (vq_prepare) and the packed conv weights an EAGER forward cached between replays must be rebuilt.  Scenario: replays,
    an eval encode + discriminator forward (fills the caches from the then-current weights), MORE replays, then the same eval
    calls -- codes must be the exact argmin over the codebook as it is NOW (quantize2_mask.py:117-128 rewrites the weight every
    training forward), PatchGAN logits must come from the current discriminator weights."""
    from dynamicvectorquantization_amd import runtime as rt
    from oracle import losses as olo
    from oracle import vq as ovq
    xs = [torch.from_numpy(synth.half_flat_images(2, 64, seed=40 + i)).to(dev) for i in range(3)]
    xv = torch.from_numpy(synth.half_flat_images(2, 64, seed=77)).to(dev)
    with rt.compute_dtype_ctx(torch.float32):
        model, tr = _make(dev, True, "full")
        def probe():
            model.eval()
            with torch.no_grad():
                ent = model.entropy_calculation(xv)
                hd = model.encoder(xv, ent)
                h = model.quant_conv(hd[model.encoder.OUT_KEY])                   # [B,D,H,W]
                _, _, info, grain, _, _ = model.encode(xv)
                logits = model.loss.discriminator(xv.contiguous())
            model.train()
            torch.cuda.synchronize()
            return h.float().cpu().numpy(), info[2].cpu().numpy(), grain.cpu().numpy(), logits.float().cpu()

        def expect(h):
            cb = model.quantize.codebook.weight.detach()[:-1].float().cpu().numpy()
            b, d, hh, ww = h.shape
            idx = ovq.argmin_exact(np.ascontiguousarray(h.transpose(0, 2, 3, 1)).reshape(-1, d), cb)
            return idx.reshape(b, hh, ww), cb

        step = 0
        for _ in range(5):                                   # 2 eager + recording + 3 replays
            tr.train_step({"image": xs[step % 3]}, step)
            step += 1
        assert tr.graph_replays >= 3
        h0, codes0, _, logits0 = probe()                     # fills the eager-side caches
        idx0, cb0 = expect(h0)
        np.testing.assert_array_equal(codes0, idx0)
        r0 = tr.graph_replays
        for _ in range(3):                                   # replays only: no Python-side invalidation but the Trainer's
            tr.train_step({"image": xs[step % 3]}, step)
            step += 1
        assert tr.graph_replays == r0 + 3
        h1, codes1, _, logits1 = probe()
        idx1, cb1 = expect(h1)
        assert np.abs(cb1 - cb0).max() > 0                   # the codebook did move under the replays
        np.testing.assert_array_equal(codes1, idx1)
        # discriminator (eval mode: BatchNorm running statistics, which the replays advanced too) from the CURRENT state_dict
        sd = {k[len("loss.discriminator."):]: v.detach().float().cpu() for k, v in model.state_dict().items()
              if k.startswith("loss.discriminator.")}
        want = olo.patchgan(sd, xv.float().cpu(), train=False)
        assert float((logits1 - want).norm()) <= 2e-3 * float(want.norm()) + 1e-5
        assert float((logits1 - logits0).norm()) > 1e-4 * float(logits0.norm())      # and they did change
