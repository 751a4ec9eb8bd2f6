"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol of the header, the
plugin boundary resolves the reference's dotted paths, state_dict layouts match the reference, the product
path refuses to run without a GPU, and the data-parallel pieces work over gloo (world size 2)."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_golden


def test_library_exports_every_header_symbol():
    from dynamicvectorquantization_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "dvq_hip.h")).read()
    names = sorted(set(re.findall(r"\b(dvq_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dvq_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.dvq_version() >= 100
    assert lib.dvq_vq_prep_bytes(1024, 256) == 256 + 4096 + 2 * 1024 * 256 * 2 + 4096        # header, |e|^2, two bf16 planes, |e| (round 5)
    assert lib.dvq_vq_argmin_workspace_bytes(65536) >= 4 * 65536


def test_error_convention_without_gpu():
    from dynamicvectorquantization_amd import _lib
    lib = _lib.load()
    rc = lib.dvq_vq_argmin(None, 0, None, None, 1, 1, 1, None, None, 0, None)
    assert rc == -1 and b"null" in lib.dvq_last_error()
    with pytest.raises(_lib.DvqError):
        _lib.check(rc, "dvq_vq_argmin")


def test_plugin_boundary_and_config():
    from dynamicvectorquantization_amd import config as cfg
    with pytest.raises(KeyError):
        cfg.instantiate_from_config({"params": {}})
    vq = cfg.instantiate_from_config({"target": "modules.vector_quantization.quantize2_mask.VectorQuantize2",
                                      "params": dict(codebook_size=16, codebook_dim=8)})
    assert vq.codebook.weight.shape == (17, 8) and not vq.codebook.weight.requires_grad
    assert float(vq.codebook.weight.abs().max()) <= 1 / 16
    cfg.install_reference_aliases()
    from modules.dynamic_modules.EncoderDual import DualGrainEncoder  # noqa: F401
    from modules.diffusionmodules.model import ResnetBlock  # noqa: F401
    from utils.utils import instantiate_from_config  # noqa: F401
    base = cfg.load_yaml(os.path.join(REPO, "configs/stage1/dqvae-entropy-dual-r05_imagenet.yml"))
    over = cfg.from_dotlist(["model.params.image_size=64", "model.params.encoderconfig.params.resolution=64",
                             "model.params.lossconfig.params.perceptual_weight=0.0", "data.params.batch_size=2"])
    merged = cfg.merge(base, over)
    assert merged.model.params.image_size == 64 and merged.model.params.encoderconfig.params.ch == 128
    assert merged.model.params.vqconfig.params.codebook_size == 1024 and merged.data.params.batch_size == 2
    assert merged.model.target == "models.stage1_dynamic.dqvae_dual_entropy.DualGrainVQModel"
    r = cfg.instantiate_from_config({"target": "modules.dynamic_modules.RouterDual.DualGrainFeatureRouter",
                                     "params": dict(num_channels=64, normalization_type="group-32", gate_type="2layer-fc-SiLu")})
    assert r.gate[0].weight.shape == (128, 128) and r.gate[2].weight.shape == (2, 128) and r.feature_norm_fine.weight.shape == (64,)
    with pytest.raises(NotImplementedError):      # same exception type as RouterDual.py:20
        cfg.instantiate_from_config({"target": "modules.dynamic_modules.RouterDual.DualGrainFeatureRouter",
                                     "params": dict(num_channels=64, gate_type="3layer")})
    t3 = cfg.instantiate_from_config({"target": "modules.dynamic_modules.RouterTriple.TripleGrainFeatureRouter",
                                      "params": dict(num_channels=32, normalization_type="none", gate_type="1layer-fc")})
    assert t3.gate.weight.shape == (3, 96)


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_state_dict_layout_matches_reference(tag):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_model import GEOM, model_config
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden(f"dqvae_{tag}")
    model = instantiate_from_config(model_config(**GEOM[tag]))
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(g["state_keys"], g["state_shapes"])}
    assert own == ref, (set(own) ^ set(ref))


def test_router_threshold_key_arithmetic():
    from dynamicvectorquantization_amd.dqvae import DualGrainFixedEntropyRouter
    import json
    path = os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json")
    tab = json.load(open(path))
    for r, key in ((0.5, "50"), (0.3, "70"), (0.55, "44"), (0.56, "43"), (0.7, "30")):
        assert DualGrainFixedEntropyRouter(path, r).fine_grain_threshold == tab[key]
    rt_ = DualGrainFixedEntropyRouter(path, 0.5)
    ent = torch.tensor([[[0.5, 1.6777750253677368], [1.68, 3.0]]])
    gate = rt_(entropy=ent)
    assert gate.tolist() == [[[[1, 0], [1, 0]], [[0, 1], [0, 1]]]]


def test_no_cpu_fallback():
    from dynamicvectorquantization_amd import _lib
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    vq = VectorQuantize2(codebook_size=16, codebook_dim=64).eval()
    with pytest.raises(_lib.DvqError):
        vq(torch.zeros(1, 64, 2, 2))


def test_schedules_losses_match_reference():
    from dynamicvectorquantization_amd import losses as L
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.trainer import scheduler_linear_warmup, scheduler_linear_warmup_cosine_decay
    g = load_golden("losses")
    f1 = scheduler_linear_warmup_cosine_decay(10, 100, 0.01)
    f2 = scheduler_linear_warmup(7)
    np.testing.assert_allclose([f1(s) for s in range(120)], g["sched_cos"], rtol=0, atol=0)
    np.testing.assert_allclose([f2(s) for s in range(20)], g["sched_lin"], rtol=0, atol=0)
    lr = np.array([synth.det_param("loss.lr", (64,)), synth.det_param("loss.lf", (64,))]) * 30
    np.testing.assert_allclose(L.hinge_d_loss(torch.from_numpy(lr[0]), torch.from_numpy(lr[1])).item(), g["hinge_d"], rtol=1e-6)
    np.testing.assert_allclose(L.hinge_g_loss(torch.from_numpy(lr[1])).item(), g["hinge_g"], rtol=1e-6)
    gate = (synth.det_param("budget.gate", (4, 2, 16, 16)) > 0).astype(np.float32)
    gate[:, 1] = 1 - gate[:, 0]
    for ca in (True, False):
        bl = L.BudgetConstraint_RatioMSE_DualGrain(target_ratio=0.5, gamma=10.0, min_grain_size=16, max_grain_size=32, calculate_all=ca)
        np.testing.assert_allclose(bl(torch.from_numpy(gate)).item(), g[f"budget_dual_{int(ca)}"], rtol=1e-5)
    bl3 = L.BudgetConstraint_NormedSeperateRatioMSE_TripleGrain(target_fine_ratio=0.3, target_median_ratio=0.3, gamma=10.0,
                                                                 min_grain_size=8, median_grain_size=16, max_grain_size=32)
    np.testing.assert_allclose(bl3(torch.from_numpy(g["budget_triple_gate"])).item(), g["budget_triple"], rtol=1e-5)


# ---- world-size-2 gloo: gradient buckets + fused VQ-EMA exchange -------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamicvectorquantization_amd.quantize import VQEmbedding
    from dynamicvectorquantization_amd.trainer import GradBuckets
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (1000, 33), (2, 2, 2, 2))]
    w0 = [p.detach().clone() for p in params]
    gb = GradBuckets(params, bucket_bytes=40000)
    assert len(gb.flat) >= 2
    for i, p in enumerate(params):
        assert torch.equal(p.detach(), w0[i])         # flattening keeps the values, .data/.grad become views
        assert p.grad.data_ptr() >= gb.fp.flat_g.data_ptr()
        p.grad.add_(float(rank + 1) * (i + 1))        # kernels accumulate in place into bucket views
    lo, hi = gb.param_range(params[:2])
    assert (lo, hi) == (0, 26)
    gb.reduce_range(hi, gb.fp.flat_g.numel())          # "decoder side" first, from inside the backward ...
    for w in gb.reduce():                              # ... the rest afterwards; every element averaged exactly once
        w.wait()
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
    # VQ-EMA exchange (quantize2_mask.py:86-88 + 99-100) as ONE all-reduce of the flat [K, D+1 | K, D] buffer: statistics are
    # summed, the restart rows every rank ends up with are rank 0's (the other ranks contribute exact zeros)
    flat, stats, restart = VQEmbedding._exchange_buffers(8, 4, "cpu")
    stats.fill_(float(rank + 1))
    r0 = torch.arange(32, dtype=torch.float32).view(8, 4) * 0.37 - 3.0
    restart.copy_(r0 if rank == 0 else torch.zeros(8, 4))
    VQEmbedding._exchange(flat)
    ok = ok and torch.equal(stats, torch.full((8, 5), 3.0)) and torch.equal(restart, r0)
    VQEmbedding._exchange(stats)                        # restart_unused_codes=False: statistics only
    ok = ok and torch.equal(stats, torch.full((8, 5), 6.0)) and torch.equal(restart, r0)
    gb.zero()
    ok = ok and all(float(p.grad.abs().sum()) == 0 for p in params)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_pieces_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("kind", ["triple", "dualfeat"])
def test_feature_routed_state_dict_layout_matches_reference(kind):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_featrouted import feat_model_config
    from dynamicvectorquantization_amd.config import instantiate_from_config
    g = load_golden(f"featrouted_{kind}")
    model = instantiate_from_config(feat_model_config(kind))
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(g["state_keys"], g["state_shapes"])}
    assert own == ref, (set(own) ^ set(ref))
    # the shipped YAMLs of the feature-routed models resolve through the plugin boundary
    from dynamicvectorquantization_amd import config as cfg
    name = "dqvae-triple-r-03-03_imagenet.yml" if kind == "triple" else "dqvae-dual-r-05_imagenet.yml"
    c = cfg.load_yaml(os.path.join(REPO, "configs/stage1", name))
    assert c.model.params.lossconfig.params.budget_loss_config.target.startswith("modules.dynamic_modules.budget.")


# ---- n1: reference (Lightning-format) checkpoints <-> this repo's classes --------------------------------------------------------
CKPT_CONFIGS = ["dqvae-entropy-dual-r05_imagenet", "dqvae-dual-r-05_imagenet", "dqvae-triple-r-03-03_imagenet"]


def _ref_layout(name):
    g = load_golden("ckpt_layout")
    keys = [str(k) for k in g[name + ".keys"]]
    shapes = [tuple(int(v) for v in str(s).split(",")) if str(s) else () for s in g[name + ".shapes"]]
    dtypes = [getattr(torch, str(d)) for d in g[name + ".dtypes"]]
    return keys, shapes, dtypes


@pytest.mark.parametrize("name", CKPT_CONFIGS)
def test_shipped_yaml_state_dict_equals_reference_layout(name):
    """the model built from the shipped YAML has the reference's complete state_dict -- same keys IN THE SAME ORDER, shapes and
    dtypes, including loss.discriminator.* and loss.perceptual_loss.* (fixture: the reference model built from ITS yaml)"""
    import warnings
    from dynamicvectorquantization_amd import config as cfg
    c = cfg.load_yaml(os.path.join(REPO, "configs", "stage1", name + ".yml"))
    os.chdir(REPO)                 # the YAMLs name the threshold table relative to the repo root, like the reference
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = cfg.instantiate_from_config(c.model)
    sd = model.state_dict()
    keys, shapes, dtypes = _ref_layout(name)
    assert list(sd.keys()) == keys, [k for k in keys if k not in sd][:5] + [k for k in sd if k not in keys][:5]
    for k, sh, dt_ in zip(keys, shapes, dtypes):
        assert tuple(sd[k].shape) == sh and sd[k].dtype == dt_, (k, tuple(sd[k].shape), sh, sd[k].dtype, dt_)


def test_lightning_checkpoint_round_trip(tmp_path):
    """a reference-produced last.ckpt ({"state_dict", "optimizer_states" in torch format, ...}) restores into the repo's model
    (init_from_ckpt with ignore_keys, dqvae_dual_entropy.py:113-122) and into the Trainer (`-r`), and what the Trainer saves has
    the reference's key list again -- so the reference classes can load it"""
    import warnings
    from dynamicvectorquantization_amd import config as cfg
    from dynamicvectorquantization_amd.trainer import Trainer
    name = CKPT_CONFIGS[0]
    keys, shapes, dtypes = _ref_layout(name)
    gen = torch.Generator().manual_seed(0)
    ref_sd = {}
    for k, sh, dt_ in zip(keys, shapes, dtypes):
        ref_sd[k] = torch.randn(sh, generator=gen).to(dt_) if dt_.is_floating_point else torch.full(sh, 7, dtype=dt_)
    path = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": ref_sd, "global_step": 11, "epoch": 0, "pytorch-lightning_version": "1.5.6"}, path)
    os.chdir(REPO)
    c = cfg.load_yaml(os.path.join(REPO, "configs", "stage1", name + ".yml"))
    c.model.params["ckpt_path"] = path
    c.model.params["ignore_keys"] = ["loss.discriminator"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = cfg.instantiate_from_config(c.model)
    sd = model.state_dict()
    for k in keys:
        if k.startswith("loss.discriminator"):
            if sd[k].dtype.is_floating_point and sd[k].numel() > 8:
                assert not torch.equal(sd[k], ref_sd[k]), k          # dropped by ignore_keys: keeps its own initialisation
        else:
            assert torch.equal(sd[k], ref_sd[k]), k
    # Trainer round trip: torch-format optimizer states of a reference run restore the moments; the saved file has the
    # reference's keys in the reference's order
    model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-4, 100, 10
    tr = Trainer(model, max_steps=1, use_graph=False)
    ae_params = model.ae_parameters()
    st = {"state": {}, "param_groups": [{"lr": 1e-4, "betas": (0.5, 0.9), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                                         "params": list(range(len(ae_params)))}]}
    for i, p in enumerate(ae_params):
        if p.requires_grad:
            st["state"][i] = {"step": torch.tensor(11.0), "exp_avg": torch.full_like(p, 0.5), "exp_avg_sq": torch.full_like(p, 0.25)}
    disc_params = list(model.loss.discriminator.parameters())
    st_d = {"state": {i: {"step": torch.tensor(11.0), "exp_avg": torch.full_like(p, 0.125), "exp_avg_sq": torch.full_like(p, 0.0625)}
                      for i, p in enumerate(disc_params)},
            "param_groups": [{"lr": 1e-4, "betas": (0.5, 0.9), "eps": 1e-8, "weight_decay": 0, "params": list(range(len(disc_params)))}]}
    ck = {"state_dict": ref_sd, "global_step": 11, "optimizer_states": [st, st_d],
          "lr_schedulers": [s["scheduler"].state_dict() for s in tr.scheds]}
    tr.load_state_dict(ck)
    assert model.global_step == 11 and tr.opts[0]._fstate["step"] == 11
    assert float(tr.opts[0]._fstate["m"].min()) == 0.5 == float(tr.opts[0]._fstate["m"].max())
    assert float(tr.opts[1]._fstate["v"].min()) == 0.0625
    # conv weights live channel-last in the flat buffer: the moments must land through the same permuted view
    w = model.encoder.conv_in.weight
    assert torch.equal(w.detach(), ref_sd["encoder.conv_in.weight"]) and not w.is_contiguous()
    saved = tr.state_dict()
    assert list(saved["state_dict"].keys()) == keys
    for k in keys:
        assert torch.equal(saved["state_dict"][k].cpu(), ref_sd[k]) and saved["state_dict"][k].shape == ref_sd[k].shape, k
    back = saved["optimizer_states"][0]
    assert len(back["state"]) == len(st["state"]) and float(back["state"][0]["step"]) == 11
    assert tuple(back["state"][0]["exp_avg"].shape) == tuple(ae_params[0].shape) and back["state"][0]["exp_avg"].is_contiguous()
    # a checkpoint whose optimizer list does not match (e.g. saved with disc_factor = 0) keeps the weights and says so
    with pytest.warns(UserWarning, match="optimizer states"):
        tr.load_state_dict({"state_dict": ref_sd, "global_step": 3, "optimizer_states": [st]})
    assert model.global_step == 3


def test_lpips_pretrained_loader(tmp_path, monkeypatch):
    """LPIPS must be loadable from torchvision's VGG16 file + the LPIPS lin file, and training with
    perceptual_weight > 0 on random features must warn loudly"""
    from dynamicvectorquantization_amd import losses
    sd = {}
    for _, idxs, chans in losses.vgg16.SLICES:
        for j, i in enumerate(idxs):
            sd[f"features.{i}.weight"] = torch.randn(chans[j + 1], chans[j], 3, 3)
            sd[f"features.{i}.bias"] = torch.randn(chans[j + 1])
    lin = {f"lin{k}.model.1.weight": torch.rand(1, c, 1, 1) for k, c in enumerate([64, 128, 256, 512, 512])}
    torch.save(sd, tmp_path / "vgg16.pth")
    torch.save(lin, tmp_path / "lin.pth")
    monkeypatch.delenv(losses.LPIPS.ENV_VGG, raising=False)
    monkeypatch.delenv(losses.LPIPS.ENV_LIN, raising=False)
    monkeypatch.chdir(tmp_path)          # no modules/lpips/vgg.pth here
    disc = {"target": "modules.discriminator.model.NLayerDiscriminator", "params": dict(input_nc=3, ndf=8, n_layers=3)}
    with pytest.warns(UserWarning, match="RANDOM frozen VGG16"):
        m = losses.VQLPIPSWithDiscriminator(disc_start=0, disc_config=disc, disc_init=True, perceptual_weight=1.0)
    assert not m.perceptual_loss.pretrained_loaded
    monkeypatch.setenv(losses.LPIPS.ENV_VGG, str(tmp_path / "vgg16.pth"))
    monkeypatch.setenv(losses.LPIPS.ENV_LIN, str(tmp_path / "lin.pth"))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")       # no warning once the weights are there
        m = losses.VQLPIPSWithDiscriminator(disc_start=0, disc_config=disc, disc_init=True, perceptual_weight=1.0)
    lp = m.perceptual_loss
    assert lp.pretrained_loaded and torch.equal(lp.net.slice4[2].weight, sd["features.21.weight"])
    assert torch.equal(lp.lin3.model[-1].weight, lin["lin3.model.1.weight"]) and not lp.lin3.model[-1].weight.requires_grad
    sd["features.0.weight"] = torch.randn(64, 4, 3, 3)
    torch.save(sd, tmp_path / "bad.pth")
    with pytest.raises(ValueError):
        lp.load_pretrained(str(tmp_path / "bad.pth"), str(tmp_path / "lin.pth"))


def test_train_py_gpu_id_mapping():
    sys.path.insert(0, REPO)
    import train
    assert train._gpu_ids("-1", 8) == list(range(8)) and train._gpu_ids("3", 8) == [0, 1, 2]
    assert train._gpu_ids("2,3", 8) == [2, 3] and train._gpu_ids("5,", 8) == [5]
