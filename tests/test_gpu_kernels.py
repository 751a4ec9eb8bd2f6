"""GPU parity tests: every HIP kernel family against the CPU oracle / reference goldens, through the
C ABI (ctypes) exactly as the product path calls it.  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, load_golden
from dynamicvectorquantization_amd import synth

pytestmark = pytest.mark.gpu

THR_JSON = os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_{}_patch-16.json")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


# ---------------------------------------------------------------------------------------------
# VQ argmin: bit-exact indices
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ci", range(5))
@pytest.mark.parametrize("impl", [0, 1])
def test_vq_argmin_golden(dev, ci, impl):
    from dynamicvectorquantization_amd import kernels as K
    g = load_golden("vq_argmin")
    n, d, k, seed = (int(v) for v in g[f"case{ci}_meta"])
    x, cb = synth.vq_inputs(n, d, k, str(g[f"case{ci}_dist"]), seed)
    idx, flagged = K.vq_argmin(T(x, dev), T(cb, dev), impl=impl, return_flagged=True)
    idx = idx.cpu().numpy()
    exact = g[f"case{ci}_exact_idx"]
    assert np.array_equal(idx, exact), f"{(idx != exact).sum()} / {n} rows differ from the exact argmin"
    # ... and versus the REFERENCE's own CPU indices (VQEmbedding.find_nearest_embedding, fp32 addmm): the device result may
    # differ only where the reference's answer is fp32 rounding noise -- at most 2 rows per case, each a near tie whose
    # exact best / second-best gap (recorded in the fixture) is below 4e-7 * (|x|^2 + 1)
    ref = g[f"case{ci}_ref_idx"]
    bad = np.nonzero(idx != ref)[0]
    assert len(bad) <= 2, f"{len(bad)} rows differ from the reference indices"
    scale = (x[bad].astype(np.float64) ** 2).sum(axis=1) + 1.0
    assert np.all(g[f"case{ci}_gap"][bad] < 4e-7 * scale), (g[f"case{ci}_gap"][bad], scale)
    if impl == 0 and d in (64, 128, 256):
        frac = int(flagged[:2].sum().item()) / n
        assert frac < 0.2, f"fp64 re-rank fraction {frac} unexpectedly high"


def test_vq_argmin_ties(dev):
    from dynamicvectorquantization_amd import kernels as K
    g = load_golden("vq_argmin")
    for impl in (0, 1):
        idx = K.vq_argmin(T(g["tie_x"], dev), T(g["tie_cb"], dev), impl=impl).cpu().numpy()
        assert np.array_equal(idx, g["tie_exact_idx"])
        # exact ties are where the reference (first minimum of its fp32 distances) is BLAS-noise dependent: at most one row
        assert (idx != g["tie_ref_idx"]).sum() <= 1
    # duplicated codes on the MFMA path (D = 64)
    rs = np.random.RandomState(5)
    cb = rs.standard_normal((96, 64)).astype(np.float32)
    cb[50] = cb[3]
    cb[95] = cb[3]
    cb[67] = cb[3]          # 64 apart: same residue class mod 32 as code 3 -> two candidates in one lane (full re-rank path)
    cb[40] = cb[8] + np.float32(1e-7)   # near-duplicate in the same class
    x = rs.standard_normal((300, 64)).astype(np.float32)
    x[:10] = cb[3] + 1e-3 * rs.standard_normal((10, 64)).astype(np.float32)
    from oracle import vq as ovq
    exact = ovq.argmin_exact(x, cb)
    x[10:20] = cb[8] + 1e-3 * rs.standard_normal((10, 64)).astype(np.float32)
    exact = ovq.argmin_exact(x, cb)
    idx, flagged = K.vq_argmin(T(x, dev), T(cb, dev), impl=2, return_flagged=True)
    idx = idx.cpu().numpy()
    assert np.array_equal(idx, exact)
    assert np.all(idx[:10] == 3)
    assert int(flagged[:2].sum()) >= 20         # the rows nearest to the duplicated codes were re-ranked (same-class duplicates: whole class)


@pytest.mark.parametrize("d", [64, 256])
@pytest.mark.parametrize("rows", ["fp32", "bf16"])
def test_vq_argmin_per_code_bound_mixed_norms(dev, d, rows):
    """the selection's error bound uses each code's OWN norm (round 5): codebooks whose norms span four orders of magnitude, rows placed
    a hair off the bisector of a short and a long code, of two long codes and of same-class pairs, plus rows at every scale -- the
    result must stay the exact argmin (fp64 oracle), and the bound must flag far fewer rows than one global max-norm bound would"""
    from dynamicvectorquantization_amd import kernels as K
    from oracle import vq as ovq
    rs = np.random.RandomState(11 + d)
    k = 1024
    cb = rs.standard_normal((k, d)).astype(np.float32)
    scale = np.exp(rs.uniform(np.log(0.01), np.log(60.0), size=(k, 1))).astype(np.float32)
    cb = cb / np.linalg.norm(cb, axis=1, keepdims=True) * scale
    xs = []
    pairs = [(rs.randint(k), rs.randint(k)) for _ in range(1500)] + [(j, j + 32 * rs.randint(1, 8)) for j in rs.randint(0, k - 256, 500)]
    for a, b in pairs:                       # near-bisector rows: exact top-2 gap ~1e-6 .. 1e-3 of the scores
        if a == b:
            continue
        mid = 0.5 * (cb[a] + cb[b])
        dirn = (cb[b] - cb[a]) / max(1e-20, np.linalg.norm(cb[b] - cb[a]))
        eps = 10.0 ** rs.uniform(-7, -3) * max(np.linalg.norm(cb[a]), np.linalg.norm(cb[b]))
        xs.append(mid + (eps if rs.rand() < 0.5 else -eps) * dirn)
    xs = np.stack(xs).astype(np.float32)
    bulk = rs.standard_normal((6000, d)).astype(np.float32) * np.exp(rs.uniform(np.log(0.01), np.log(30.0), size=(6000, 1))).astype(np.float32)
    x = np.concatenate([xs, bulk, cb[:200] * np.float32(1.0000001)], 0)
    xt = T(x, dev)
    if rows == "bf16":
        xt = xt.to(torch.bfloat16)
        x = xt.float().cpu().numpy()
    idx, flagged = K.vq_argmin(xt, T(cb, dev), impl=2, return_flagged=True)
    assert np.array_equal(idx.cpu().numpy(), ovq.argmin_exact(x, cb))
    # how many rows ONE global bound (max norm ~60 for every pair) would have sent to the fp64 re-rank: most of the short-code rows
    n_amb = int(flagged[:2].sum())
    assert n_amb < 0.6 * len(x), (n_amb, len(x))


@pytest.mark.parametrize("k", [1024, 8192])
def test_vq_argmin_full_size_properties(dev, k):
    """BASELINE size N=65536: MFMA path == fp64 path on a row sample; codebook rows map to themselves."""
    from dynamicvectorquantization_amd import kernels as K
    from oracle import vq as ovq
    n, d = 65536, 256
    for dist_ in ("normal", "encoder"):
        x, cb = synth.vq_inputs(n, d, k, dist_, 0)
        xt, cbt = T(x, dev), T(cb, dev)
        idx, flagged = K.vq_argmin(xt, cbt, impl=2, return_flagged=True)
        idx = idx.cpu().numpy()
        sample = np.random.RandomState(1).choice(n, 2048, replace=False)
        assert np.array_equal(idx[sample], ovq.argmin_exact(x[sample], cb))
        assert int(flagged[:2].sum().item()) < 0.1 * n
        # ambiguous rows are settled among their few candidate codes (+ whole residue classes); the all-codes re-rank (more
        # than VQ_MAXC candidate classes) is the rare exception
        assert int(flagged[0]) == 0 and int(flagged[2]) <= 0.05 * int(flagged[1]) + 4, flagged.cpu().numpy()
        # idempotence: quantising code vectors returns their own index
        self_idx = K.vq_argmin(cbt, cbt, impl=2).cpu().numpy()
        assert np.array_equal(self_idx, np.arange(k))
        # bf16 activations (perf mode): still exact w.r.t. the bf16-rounded rows
        xb = xt[:4096].to(torch.bfloat16)
        idxb = K.vq_argmin(xb, cbt, impl=2).cpu().numpy()
        assert np.array_equal(idxb, ovq.argmin_exact(xb.float().cpu().numpy(), cb))


@pytest.mark.parametrize("d", [64, 128, 256])
def test_vq_argmin_bf16_rows_kernel(dev, d):
    """bf16 rows (the training path) run on their own main kernel (4 waves x 64 rows, bookkeeping pipelined under the MFMAs):
    exact argmin w.r.t. the bf16-rounded rows for every supported D, ragged row counts (tile tails, fewer rows than a tile),
    K not a multiple of 32, both benchmark distributions, exact many-way ties (more candidate classes than an entry lists ->
    class scans, no all-codes row) -- and repeated calls on the shared scratch, whose counters the re-rank kernel re-arms."""
    from dynamicvectorquantization_amd import kernels as K
    from oracle import vq as ovq
    for n, k, dist_, seed in ((4096, 1024, "normal", 1), (4096 + 77, 1000, "encoder", 2), (130, 96, "normal", 3), (7, 33, "normal", 4)):
        x, cb = synth.vq_inputs(n, d, k, dist_, seed)
        xb = T(x, dev).to(torch.bfloat16)
        cbt = T(cb, dev)
        want = ovq.argmin_exact(xb.float().cpu().numpy(), cb)
        for rep in range(3):                               # same scratch three times: counters re-armed, reports per call
            idx, flagged = K.vq_argmin(xb, cbt, impl=2, return_flagged=True)
            assert np.array_equal(idx.cpu().numpy(), want), (n, k, dist_, rep)
            f = flagged.cpu().numpy()
            assert f[0] == 0 and 0 <= f[2] <= f[1] <= n, f
            if rep:
                assert np.array_equal(f, f_prev)           # deterministic candidate sets
            f_prev = f
    # many-way exact ties: 40 copies of one code spread over > 5 residue classes; rows next to it see > VQ_MAXC candidate classes
    rs = np.random.RandomState(11)
    k = 256
    cb = rs.standard_normal((k, d)).astype(np.float32)
    dup = rs.choice(np.arange(1, k), 40, replace=False)
    cb[dup] = cb[0]
    x = rs.standard_normal((500, d)).astype(np.float32)
    x[:64] = cb[0] + 1e-3 * rs.standard_normal((64, d)).astype(np.float32)
    xb = T(x, dev).to(torch.bfloat16)
    want = ovq.argmin_exact(xb.float().cpu().numpy(), cb)
    idx, flagged = K.vq_argmin(xb, T(cb, dev), impl=2, return_flagged=True)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.all(idx.cpu().numpy()[:64] == 0)             # lowest index among the exact ties
    f = flagged.cpu().numpy()
    assert f[0] == 0 and f[2] >= 64 and f[1] >= f[2], f    # the tied rows took the wide (class-scan) form, none an all-codes row
    # fp32 rows through the same call sequence (8-wave kernel): the wide form as well
    idx32, flagged32 = K.vq_argmin(T(x, dev), T(cb, dev), impl=2, return_flagged=True)
    assert np.array_equal(idx32.cpu().numpy(), ovq.argmin_exact(x, cb))
    assert int(flagged32[0]) == 0 and int(flagged32[2]) >= 64


def test_vq_distances_and_soft_codes_golden(dev):
    """the analysis entry points of SURVEY 8(b): VQEmbedding.compute_distances and VectorQuantize2.get_soft_codes vs the reference"""
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    g = load_golden("vq_distances")
    k, d = (int(v) for v in g["meta"])
    vq = VectorQuantize2(codebook_size=k, codebook_dim=d).to(dev).eval()
    with torch.no_grad():
        vq.codebook.weight.copy_(T(synth.det_param("vqdist.codebook", (k + 1, d)) * 4.0, dev))
    x = T(synth.det_param("vqdist.x", (2, 5, 3, d)) * 6.0, dev)
    dist = vq.codebook.compute_distances(x)
    assert tuple(dist.shape) == g["distances"].shape and dist.dtype == torch.float32
    np.testing.assert_allclose(dist.cpu().numpy(), g["distances"], rtol=2e-5, atol=2e-4)
    for temp in (1.0, 0.25):
        soft, code = vq.get_soft_codes(x, temp=temp, stochastic=False)
        np.testing.assert_allclose(soft.cpu().numpy(), g[f"soft_{temp}"], rtol=2e-3, atol=1e-6)
        assert np.array_equal(code.cpu().numpy(), g["code"])
    soft, code = vq.get_soft_codes(x, temp=1.0, stochastic=True)          # multinomial draw: device RNG, only its support is checked
    assert tuple(code.shape) == (2, 5, 3) and bool((soft.gather(-1, code.unsqueeze(-1)) > 0).all())
    # bf16 rows (perf mode) and a row count that is not a multiple of the tile
    xb = x.reshape(-1, d)[:29].to(torch.bfloat16).contiguous()
    db = vq.codebook.compute_distances(xb)
    ref = ((xb.float()[:, None, :] - vq.codebook.weight[:-1][None]) ** 2).sum(-1)
    np.testing.assert_allclose(db.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_vq_forward_golden(dev):
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    g = load_golden("vq_forward")
    for tag in ("a", "b"):
        b, d, h, k, use_mask = (int(v) for v in g[f"{tag}_meta"])
        vq = VectorQuantize2(codebook_size=k, codebook_dim=d).to(dev).eval()
        w = synth.det_param(f"vqfwd.{tag}.codebook", (k + 1, d)) * 4.0
        x = synth.det_param(f"vqfwd.{tag}.x", (b, d, h, h)) * 6.0
        with torch.no_grad():
            vq.codebook.weight.copy_(T(w, dev))
        xt = T(x, dev).requires_grad_(True)
        mask = T(g[f"{tag}_mask"], dev) if use_mask else None
        xq, loss, (_, _, idx) = vq(xt, codebook_mask=mask)
        gout = T(synth.det_param(f"vqfwd.{tag}.gout", x.shape), dev)
        (loss * 3.0 + (xq * gout).sum()).backward()
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_idx"])
        np.testing.assert_allclose(xq.detach().cpu().numpy(), g[f"{tag}_x_q"], atol=1e-6)
        np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=1e-5)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), g[f"{tag}_dx"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(vq.get_codebook_entry(idx).cpu().numpy(), g[f"{tag}_entry"])


def test_vq_ema_golden(dev):
    from dynamicvectorquantization_amd.quantize import VQEmbedding
    g = load_golden("vq_ema")
    for tag, dead in (("live", False), ("dead", True)):
        n, d, k = (int(v) for v in g[f"{tag}_meta"])
        emb = VQEmbedding(k, d).to(dev).train()
        w = synth.det_param(f"ema.{tag}.w", (k + 1, d)) * 3.0
        x = synth.det_param(f"ema.{tag}.x", (n, d)) * (1.5 if dead else 3.0)
        n0 = g[f"{tag}_n_ema0"]
        with torch.no_grad():
            emb.weight.copy_(T(w, dev))
            emb.embed_ema.copy_(T(w[:-1] * n0[:, None], dev))
            emb.cluster_size_ema.copy_(T(n0, dev))
        emb.restart_perm = T(g[f"{tag}_perm"], dev)
        embeds, idx = emb(T(x, dev).view(1, n, d))
        assert np.array_equal(idx.cpu().numpy().reshape(-1), g[f"{tag}_idx"])
        np.testing.assert_allclose(embeds.cpu().numpy().reshape(n, d), w[:-1][g[f"{tag}_idx"]])   # OLD weight
        np.testing.assert_allclose(emb.cluster_size_ema.cpu().numpy(), g[f"{tag}_n_ema"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(emb.embed_ema.cpu().numpy(), g[f"{tag}_s_ema"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(emb.weight[:-1].detach().cpu().numpy(), g[f"{tag}_weight"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(emb.weight[-1].detach().cpu().numpy(), w[-1])                    # padding row untouched


# ---------------------------------------------------------------------------------------------
# entropy + gate: H to 1e-5, gates bit-exact
# ---------------------------------------------------------------------------------------------
def test_entropy_gate_golden(dev):
    from dynamicvectorquantization_amd import kernels as K
    from oracle import entropy as oent
    g = load_golden("entropy")
    imgs = {"small": g["small_img"], "big": synth.half_flat_images(2, 256, seed=1234)}
    n_band = n_flip = n_rows = 0
    for tag, img in imgs.items():
        for table in ("imagenet_train", "imagenet_val", "ffhq_train"):
            for r in (0.3, 0.5, 0.7, 0.55):
                thr = oent.threshold_from_table(THR_JSON.format(table), r)
                ent, gate = K.patch_entropy_gate(T(img, dev), 16, thr)
                ent = ent.cpu().numpy()
                ref = g[f"{tag}_H"]
                np.testing.assert_allclose(ent, ref, rtol=2e-5, atol=1e-36)
                margin = np.abs(ref - np.float32(thr))
                safe = margin > 1e-4 * np.maximum(1.0, np.abs(ref))
                gg = gate.cpu().numpy().astype(np.int8)
                assert np.array_equal(gg[safe], g[f"{tag}_gate_{table}_{r}"][safe])
                assert safe.mean() > 0.99
                # EVERY row, in-band ones included: the device's gate must be the reference's compare (RouterDual.py:53-57,
                # fp32 `entropy > threshold`) applied to the device's OWN entropies -- H agrees with the reference to 2e-5,
                # so a row within that distance of the threshold may legitimately land on the other side, but the gate and
                # the H the same launch reports may never disagree
                assert np.array_equal(gg, oent.entropy_gate(ent, thr).astype(np.int8))
                n_band += int((~safe).sum())
                n_flip += int((gg[~safe] != g[f"{tag}_gate_{table}_{r}"][~safe]).any(-1).sum()) if (~safe).any() else 0
                n_rows += int(safe.size)
    from test_gpu_model import _report
    _report("entropy_gate_golden", rows=n_rows, in_band_rows=n_band, in_band_rows_differing_from_reference_gate=n_flip)
    assert n_flip <= n_band
    # all-underflow patch: H = 32 * eps * ln(1/eps) needs fp32 subnormals
    ent, _ = K.patch_entropy_gate(T(g["small_img"], dev), 16, None)
    assert abs(float(ent[2, 0, 0]) - 2.947293e-37) < 1e-40


# ---------------------------------------------------------------------------------------------
# GroupNorm (+swish)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,silu", [((2, 64, 8, 8), True), ((3, 128, 20, 12), True), ((2, 32, 5, 7), False),
                                        ((1, 512, 16, 16), True), ((2, 256, 33, 31), False)])
def test_groupnorm(dev, dtype, shape, silu):
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Normalize
    n, c, h, w = shape
    rs = np.random.RandomState(c + h)
    x = (rs.standard_normal(shape) * 2 + 0.5).astype(np.float32)
    gam = (1 + 0.2 * rs.standard_normal(c)).astype(np.float32)
    bet = (0.2 * rs.standard_normal(c)).astype(np.float32)
    go = rs.standard_normal(shape).astype(np.float32)
    if dtype == torch.bfloat16:
        x, go = bf16_round(x), bf16_round(go)
    xr = torch.from_numpy(x).requires_grad_(True)
    gr, br = torch.from_numpy(gam).requires_grad_(True), torch.from_numpy(bet).requires_grad_(True)
    yr = F.group_norm(xr, 32, gr, br, 1e-6)
    if silu:
        yr = yr * torch.sigmoid(yr)
    (yr * torch.from_numpy(go)).sum().backward()
    mod = Normalize(c).to(dev)
    mod.fuse_silu = silu
    with torch.no_grad():
        mod.weight.copy_(T(gam, dev))
        mod.bias.copy_(T(bet, dev))
    with rt.compute_dtype_ctx(dtype):
        xt = T(x, dev).requires_grad_(True)
        y = mod(xt)
        (y.float() * T(go, dev)).sum().backward()
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), **tol)
    np.testing.assert_allclose(xt.grad.float().cpu().numpy(), xr.grad.numpy(), **tol)
    scale = max(1.0, float(np.abs(gr.grad.numpy()).max()))
    gtol = 2e-4 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(mod.weight.grad.cpu().numpy() / scale, gr.grad.numpy() / scale, atol=gtol)
    np.testing.assert_allclose(mod.bias.grad.cpu().numpy() / scale, br.grad.numpy() / scale, atol=gtol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nonlinearity_standalone(dev, dtype):
    """layers.nonlinearity (model.py:29-31, swish) as a standalone differentiable op for reference-side callers"""
    from dynamicvectorquantization_amd.layers import nonlinearity
    x = torch.randn(3, 5, 7, 11, device=dev).to(dtype).requires_grad_(True)
    y = nonlinearity(x)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().float().cpu().requires_grad_(True)
    yr = xr * torch.sigmoid(xr)
    yr.backward(g.float().cpu())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert y.dtype == dtype and float((y.detach().float().cpu() - yr.detach()).abs().max()) < tol
    assert float((x.grad.float().cpu() - xr.grad).abs().max()) < tol * 4


# ---------------------------------------------------------------------------------------------
# convolution: fwd / dgrad / wgrad, naive (impl 1) and MFMA (impl 2)
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # cin, cout, k, kind, H, W, N
    (32, 64, 3, "same", 8, 8, 2),
    (64, 64, 3, "same", 12, 20, 2),
    (128, 128, 3, "same", 16, 16, 3),
    (64, 128, 1, "same", 9, 7, 2),
    (64, 64, 3, "down", 8, 8, 2),
    (128, 128, 3, "down", 16, 12, 2),
    (64, 64, 3, "up", 6, 6, 2),
    (128, 64, 3, "up", 8, 4, 1),
    (3, 32, 3, "same", 16, 16, 2),
    (64, 3, 3, "same", 16, 16, 2),
    (256, 256, 3, "same", 16, 16, 1),
    (16, 32, 4, "s2p1", 16, 16, 2),
    (64, 128, 4, "s2p1", 24, 16, 3),       # stride-2 input gradient by parity class (Cout % 64 == 0)
    (128, 64, 3, "down", 20, 24, 2),
]


def _conv_ref(x, w, b, kind, k):
    if kind == "same":
        return F.conv2d(x, w, b, stride=1, padding=(k - 1) // 2)
    if kind == "down":
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    if kind == "up":
        return F.conv2d(x.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2), w, b, stride=1, padding=1)
    if kind == "s2p1":
        return F.conv2d(x, w, b, stride=2, padding=1)
    raise ValueError(kind)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("impl", [1, 2])
def test_conv(dev, case, dtype, impl):
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    cin, cout, k, kind, h, w_, n = case
    rs = np.random.RandomState(cin * 7 + cout + k)
    x = rs.standard_normal((n, cin, h, w_)).astype(np.float32)
    wt = (rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    if dtype == torch.bfloat16:
        x, wt = bf16_round(x), bf16_round(wt)
    xr = torch.from_numpy(x).requires_grad_(True)
    wr, br = torch.from_numpy(wt).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yr = _conv_ref(xr, wr, br, kind, k)
    go = rs.standard_normal(tuple(yr.shape)).astype(np.float32)
    if dtype == torch.bfloat16:
        go = bf16_round(go)
    (yr * torch.from_numpy(go)).sum().backward()
    kw = dict(same=dict(stride=1, padding=(k - 1) // 2), down=dict(stride=2, padding=0, asym_pad=True),
              up=dict(stride=1, padding=1, upsample=True), s2p1=dict(stride=2, padding=1))[kind]
    mod = Conv2d(cin, cout, k, **kw).to(dev)
    with torch.no_grad():
        mod.weight.copy_(T(wt, dev))
        mod.bias.copy_(T(b, dev))
    pad_in = (cin % (4 if dtype == torch.float32 else 8)) != 0
    with rt.compute_dtype_ctx(dtype), rt.impl_ctx(impl):
        if pad_in or cout % (4 if dtype == torch.float32 else 8):
            pytest.skip("channel-padded image convs are exercised through the encoder/decoder tests")
        xt = T(x, dev).requires_grad_(True)
        y = mod(xt)
        (y.float() * T(go, dev)).sum().backward()
    if dtype == torch.float32:
        tol = dict(rtol=1e-4, atol=1e-4)
    else:
        tol = dict(rtol=2e-2, atol=2e-2 * float(np.abs(yr.detach().numpy()).max()))
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), **tol)
    gtol = 1e-4 if dtype == torch.float32 else 2e-2
    for name, got, ref in (("dx", xt.grad.float().cpu().numpy(), xr.grad.numpy()),
                           ("dw", mod.weight.grad.cpu().numpy(), wr.grad.numpy()),
                           ("db", mod.bias.grad.cpu().numpy(), br.grad.numpy())):
        s = max(1e-6, float(np.abs(ref).max()))
        err = float(np.abs(got - ref).max()) / s
        assert err < gtol * 5, f"{name}: rel-to-max error {err}"


PIPE_CASES = [
    # cin, cout, k, kind, H, W, N      -- tile of conv_nt_pipe_kernel the forward / the input gradient lands on
    (64, 128, 4, "s2p1", 24, 16, 3),       # 128 x 128 / 512 x 64 (parity classes, 64 gradient columns)
    (128, 128, 3, "down", 16, 12, 2),      # 128 x 128, stride 2 with the asymmetric pad / parity classes with 1, 2, 2, 4 taps
    (256, 256, 3, "same", 16, 16, 1),      # 16-wide maps (not halo eligible): 128 x 128
    (256, 512, 4, "s1p1", 12, 13, 2),      # 4 x 4 stride 1 (PatchGAN tail), odd output sizes, row tail inside a tile
    (512, 8, 4, "s1p1", 10, 9, 2),         # 8 output channels (PatchGAN logits)
    (64, 256, 3, "down", 128, 128, 3),     # 49152 output pixels x 256 channels: the 256 x 256 tile
    (128, 64, 1, "same", 20, 24, 2),       # 1 x 1, 64 output columns: 512 x 64
    (64, 320, 1, "same", 17, 9, 3),        # 1 x 1 with a column tail (320 = 256 + 64) and a row tail
]


@pytest.mark.parametrize("case", PIPE_CASES, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("epi", ["plain", "lrelu+gate"])
def test_conv_pipelined_igemm_kernel(dev, case, epi):
    """conv_nt_pipe_kernel (impl 9 = required) against torch's fp32 convolution: forward (+ fused LeakyReLU), input gradient
    (+ the activation gate of the layer below), with the weight gradient of the same call on the 128 x 128 TN kernel"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    cin, cout, k, kind, h, w_, n = case
    rs = np.random.RandomState(cin + 3 * cout + k)
    x = bf16_round(rs.standard_normal((n, cin, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    gate = epi == "lrelu+gate"
    kw = dict(same=dict(stride=1, padding=(k - 1) // 2), down=dict(stride=2, padding=0, asym_pad=True), s2p1=dict(stride=2, padding=1),
              s1p1=dict(stride=1, padding=1))[kind]
    mod = Conv2d(cin, cout, k, **kw).to(dev)
    with torch.no_grad():
        mod.weight.copy_(T(wt, dev))
        mod.bias.copy_(T(b, dev))
    cout_p = -(-cout // 8) * 8
    with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(9):
        x_nhwc = T(x, dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        a_in = torch.where(x_nhwc > 0, x_nhwc, (x_nhwc.float() * 0.2).to(torch.bfloat16)) if gate else x_nhwc
        tape = Tape()
        y = mod.fwd(a_in, tape, act=K.ACT_LRELU if gate else K.ACT_NONE)
    # reference: LeakyReLU is piecewise linear -- its slope pattern is taken from the device result, so that pre-activations that
    # round across zero in bf16 do not turn into O(1) gradient differences
    xr = torch.from_numpy(x).requires_grad_(True)
    xin = F.leaky_relu(xr, 0.2) if gate else xr            # the layer below: its gate is applied to dx by the kernel
    if kind == "down":
        yr = F.conv2d(F.pad(xin, (0, 1, 0, 1)), torch.from_numpy(wt), torch.from_numpy(b), stride=2)
    else:
        yr = F.conv2d(xin, torch.from_numpy(wt), torch.from_numpy(b), stride=kw["stride"], padding=kw["padding"])
    if gate:
        slope = torch.where(y[..., :cout].float().cpu().permute(0, 3, 1, 2) > 0, 1.0, 0.2)
        yr = yr * slope
    go = bf16_round(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    (yr * torch.from_numpy(go)).sum().backward()
    with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(9):
        gy = torch.zeros(n, y.shape[1], y.shape[2], cout_p, device=dev)
        gy[..., :cout] = T(go, dev).permute(0, 2, 3, 1)
        if gate:
            gy = gy * torch.where(y > 0, 1.0, 0.2)          # through the fused LeakyReLU of this layer
        if cout % 64:
            tape.s["d"].impl = 0                            # (8 gradient channels: not a 64-channel K slab -- automatic choice)
        dx = mod.bwd(gy.to(torch.bfloat16), tape, mask=a_in if gate else None, mask_act=K.ACT_LRELU if gate else K.ACT_NONE)
    ref = yr.detach().permute(0, 2, 3, 1).numpy()
    got = y.float().cpu().numpy()[..., :cout]
    err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
    assert err < 2e-2, f"forward: rel-to-max error {err}"
    dref = xr.grad.permute(0, 2, 3, 1).numpy()
    dgot = dx.float().cpu().numpy()
    derr = float(np.abs(dgot - dref).max()) / max(1e-6, float(np.abs(dref).max()))
    assert derr < 3e-2, f"input gradient: rel-to-max error {derr}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_multi_tensor_weight_pack_matches_single_pack(dev, dtype):
    """dvq_pack_weights_multi (one launch refreshing every conv's packed copies after an optimizer step) against dvq_pack_weight per layer:
    channel-padded image convs, 1x1 / 3x3 / 4x4, OIHW masters"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    torch.manual_seed(3)
    mods = [Conv2d(3, 128, 3, 1, 1), Conv2d(128, 128, 3, 1, 1), Conv2d(128, 3, 3, 1, 1), Conv2d(256, 256, 1), Conv2d(64, 128, 4, 2, 1),
            Conv2d(40, 72, 3, 1, 1), Conv2d(512, 1, 4, 1, 1)]
    mods = [m.to(dev) for m in mods]
    with rt.compute_dtype_ctx(dtype):
        for m in mods:
            m.packed(dtype)                                   # first use: single-tensor pack, registers the buffers
        with torch.no_grad():
            for m in mods:
                m.weight.add_(torch.randn_like(m.weight))     # version bump: the next packed() call repacks ALL of them in one launch
        got = [tuple(t.clone() for t in m.packed(dtype)[:2]) for m in mods]
    for m, (w, wt) in zip(mods, got):
        cin_p, cout_p = m._padded(dtype)
        rw, rwt = K.pack_weight(m.weight.detach(), cin_p, cout_p, dtype)
        assert torch.equal(w[: m.out_channels], rw), f"w differs for {tuple(m.weight.shape)}"
        assert torch.equal(wt, rwt), f"wt differs for {tuple(m.weight.shape)}"
        assert float(w[m.out_channels:].abs().sum()) == 0.0


TN_PATCH_CASES = [
    # cin, cout, k, kind, H, W, N      -- weight gradients on conv_tn_patch_kernel (8 x 8 output-pixel patches as reduction stages)
    (256, 256, 3, "same", 16, 16, 2),      # 2 x 2 patches per image, padding on every side
    (128, 128, 3, "down", 16, 12, 2),      # stride 2 + asymmetric pad, 8 x 6 outputs: a partial patch column
    (64, 128, 4, "s2p1", 24, 16, 3),       # 4 x 4 stride 2, 12 x 8 outputs: a partial patch row
    (256, 512, 4, "s1p1", 12, 13, 2),      # 4 x 4 stride 1, odd 11 x 12 outputs (PatchGAN tail), four column tiles
    (512, 8, 4, "s1p1", 10, 9, 2),         # 8 gradient rows (PatchGAN logits)
    (64, 64, 3, "up", 6, 6, 2),            # nearest x2 upsampling folded into the gather
    (128, 64, 1, "same", 20, 24, 2),       # 1 x 1 below the 256-wide GEMM kernel's size
    (64, 256, 3, "down", 40, 24, 1),       # more patches than splits
]


@pytest.mark.parametrize("case", TN_PATCH_CASES, ids=lambda c: "-".join(map(str, c)))
def test_conv_weight_gradient_patch_kernel(dev, case):
    """conv_tn_patch_kernel (automatic choice for bf16 weight gradients off the halo kernel's shapes) against torch's fp32 convolution"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.layers import Conv2d
    cin, cout, k, kind, h, w_, n = case
    rs = np.random.RandomState(11 * cin + cout + k + h)
    x = bf16_round(rs.standard_normal((n, cin, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    xr = torch.from_numpy(x)
    wr, br = torch.from_numpy(wt).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=1, padding=1) if kind == "s1p1" else _conv_ref(xr, wr, br, kind, k)
    go = bf16_round(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    (yr * torch.from_numpy(go)).sum().backward()
    kw = dict(same=dict(stride=1, padding=(k - 1) // 2), down=dict(stride=2, padding=0, asym_pad=True),
              up=dict(stride=1, padding=1, upsample=True), s2p1=dict(stride=2, padding=1), s1p1=dict(stride=1, padding=1))[kind]
    mod = Conv2d(cin, cout, k, **kw).to(dev)
    with torch.no_grad():
        mod.weight.copy_(T(wt, dev))
        mod.bias.copy_(T(b, dev))
    with rt.compute_dtype_ctx(torch.bfloat16):
        d = mod._desc(T(x, dev).permute(0, 2, 3, 1).to(torch.bfloat16))
        assert K._tn_family(d, cin) == "conv_tn_patch_kernel" or K._halo_eligible(d)
        xt = T(x, dev).requires_grad_(True)
        y = mod(xt)
        (y.float() * T(go, dev)).sum().backward()
    for name, got, ref in (("dw", mod.weight.grad.cpu().numpy(), wr.grad.numpy()), ("db", mod.bias.grad.cpu().numpy(), br.grad.numpy())):
        err = float(np.abs(got - ref).max()) / max(1e-6, float(np.abs(ref).max()))
        assert err < 1e-2, f"{name}: rel-to-max error {err}"


def test_split_bf16_planes(dev):
    """dvq_split_bf16_planes: hi = RNE(x) (torch's own fp32 -> bf16 cast), lo = RNE(x - hi); hi + lo reproduces x to ~2^-17; channels
    beyond the fp32 tensor's (4-padded -> 8-padded) are zero"""
    from dynamicvectorquantization_amd import kernels as K
    torch.manual_seed(3)
    for rows, c, cout in ((1000, 128, 128), (777, 4, 8), (513, 12, 16), (64, 64, 64)):
        x = torch.randn(rows, c, device=dev) * torch.logspace(-6, 6, rows, device=dev)[:, None]
        hi, lo = K.split_bf16_planes(x, cout)
        assert torch.equal(hi[:, :c], x.to(torch.bfloat16))
        assert torch.equal(lo[:, :c], (x - hi[:, :c].float()).to(torch.bfloat16))
        if cout > c:
            assert float(hi[:, c:].float().abs().max()) == 0.0 and float(lo[:, c:].float().abs().max()) == 0.0
        err = (hi[:, :c].float() + lo[:, :c].float() - x).abs() / x.abs().clamp_min(1e-30)
        assert float(err.max()) < 2.0 ** -16


def test_fp32_split_weight_gradient_upsampled_input(dev, monkeypatch):
    """the plane-splitting weight gradient on a convolution that reads its input through the folded nearest x2 upsample (the stored
    tensor is half-size: the planes are too), against the in-kernel split"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    torch.manual_seed(8)
    mod = Conv2d(128, 128, 3, stride=1, padding=1, upsample=True).to(dev)
    x = torch.randn(2, 128, 16, 32, device=dev)
    got = {}
    with rt.compute_dtype_ctx("fp32x3"):
        for planes in ("1", "0"):
            monkeypatch.setenv("DVQ_X3_WGRAD_PLANES", planes)
            mod.weight.grad = mod.bias.grad = None
            xt = x.clone().requires_grad_(True)
            y = mod(xt)
            assert tuple(y.shape) == (2, 128, 32, 64)
            go = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
            (y * go).sum().backward()
            got[planes] = (mod.weight.grad.double().clone(), mod.bias.grad.double().clone())
    xr = torch.nn.functional.interpolate(x.double().cpu(), scale_factor=2.0, mode="nearest")
    wr = mod.weight.detach().double().cpu().requires_grad_(True)
    br = mod.bias.detach().double().cpu().requires_grad_(True)
    (F.conv2d(xr, wr, br, padding=1) * go.double().cpu()).sum().backward()
    for planes in ("1", "0"):
        for a, r in zip(got[planes], (wr.grad, br.grad)):
            assert float((a.cpu() - r).norm() / r.norm()) < 3e-5, planes


def test_fp32_split_halo_variants(dev, monkeypatch):
    """fp32x3 forward / input gradient on the halo kernel (bf16 planes on the channel axis, fp32 output: dvq_conv2d_fwd_x3 /
    dvq_conv2d_dgrad_x3) against float64 and against the in-kernel split (DVQ_X3_HALO=0): residual, fused ReLU / LeakyReLU, gated input
    gradient, folded nearest x2 upsample, 64- / 128- / 192-channel outputs (one and two staging rounds, a partial channel block)"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    torch.manual_seed(11)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    with rt.compute_dtype_ctx("fp32x3"):
        for cin, cout, h, w_, n, up in ((128, 128, 16, 32, 2, False), (64, 64, 8, 64, 3, False), (128, 192, 8, 32, 2, False),
                                        (256, 128, 16, 32, 1, False), (128, 128, 16, 32, 2, True)):
            conv = Conv2d(cin, cout, 3, stride=1, padding=1, upsample=up).to(dev)
            x = torch.randn(n, h >> up, w_ >> up, cin, device=dev)
            d = conv._desc(x)
            monkeypatch.setenv("DVQ_X3_HALO", "1")
            assert K._x3_halo(d, x, False) and K._x3_halo(d, x, True), (cin, cout, h, w_, n, up, rt.fp32_split(), d.impl, d.H, d.W, d.OH, d.OW, d.Cin, d.Cout, d.pad_t)
            w, wt, bias = conv.packed(torch.float32)
            res = torch.randn(n, h, w_, cout, device=dev)
            dy = torch.randn(n, h, w_, cout, device=dev)
            mask = torch.randn(n, h, w_, cin, device=dev)
            xr = x.double().permute(0, 3, 1, 2)
            if up:
                xr = torch.nn.functional.interpolate(xr, scale_factor=2.0, mode="nearest")
            xr = xr.requires_grad_(True)
            yr = F.conv2d(xr, conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
            gx, = torch.autograd.grad(yr, xr, dy.double().permute(0, 3, 1, 2))
            yr = yr.detach().permute(0, 2, 3, 1)
            gx = gx.permute(0, 2, 3, 1)
            if up:
                gx = gx.reshape(n, h // 2, 2, w_ // 2, 2, cin).sum((2, 4))
            out = {}
            for halo in ("1", "0"):
                monkeypatch.setenv("DVQ_X3_HALO", halo)
                o = {"y": K.conv2d_fwd(d, x, w, bias), "y+res": K.conv2d_fwd(d, x, w, bias, res),
                     "relu": K.conv2d_fwd(d, x, w, bias, act=K.ACT_RELU), "lrelu": K.conv2d_fwd(d, x, w, bias, act=K.ACT_LRELU),
                     "dx": K.conv2d_dgrad(d, dy, wt)}
                if not up:
                    o["dx relu"] = K.conv2d_dgrad(d, dy, wt, mask, K.ACT_RELU)
                    o["dx lrelu"] = K.conv2d_dgrad(d, dy, wt, mask, K.ACT_LRELU)
                out[halo] = o
            ref = {"y": yr, "y+res": yr + res.double(), "relu": yr.clamp_min(0), "lrelu": torch.where(yr > 0, yr, 0.2 * yr), "dx": gx}
            if not up:
                ref["dx relu"] = gx * (mask > 0)
                ref["dx lrelu"] = gx * torch.where(mask > 0, 1.0, 0.2).double()
            for k, r in ref.items():
                assert rel(out["1"][k], r) < 3e-5, (cin, cout, up, k, rel(out["1"][k], r))
                assert rel(out["0"][k], r) < 3e-5, (cin, cout, up, k, "in-kernel")
            assert not torch.equal(out["1"]["y"], out["0"]["y"])          # two different kernels really ran


def test_fp32_split_halo_thin_output(dev, monkeypatch):
    """fp32x3 forward of a THIN-output 3x3 convolution (the decoder's 128 -> 3 conv_out, channels padded to 4) on the halo kernel's
    32-channel instance with the fp32 epilogue (one half-filled staging round) against float64 and the in-kernel split path"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    torch.manual_seed(12)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    with rt.compute_dtype_ctx("fp32x3"):
        for cin, cout, h, w_, n in ((128, 4, 16, 32, 2), (64, 8, 8, 64, 3), (128, 32, 8, 32, 1)):
            conv = Conv2d(cin, cout, 3, stride=1, padding=1).to(dev)
            x = torch.randn(n, h, w_, cin, device=dev)
            d = conv._desc(x)
            monkeypatch.setenv("DVQ_X3_HALO", "1")
            assert K._x3_halo(d, x, False), (cin, cout)
            w, wt, bias = conv.packed(torch.float32)
            yr = F.conv2d(x.double().permute(0, 3, 1, 2), conv.weight.detach().double(), conv.bias.detach().double(), padding=1).permute(0, 2, 3, 1)
            out = {}
            for halo in ("1", "0"):
                monkeypatch.setenv("DVQ_X3_HALO", halo)
                out[halo] = {"y": K.conv2d_fwd(d, x, w, bias), "lrelu": K.conv2d_fwd(d, x, w, bias, act=K.ACT_LRELU)}
            for k, r in (("y", yr), ("lrelu", torch.where(yr > 0, yr, 0.2 * yr))):
                assert rel(out["1"][k][..., :cout], r) < 3e-5, (cin, cout, k, rel(out["1"][k][..., :cout], r))
                assert rel(out["0"][k][..., :cout], r) < 3e-5, (cin, cout, k, "in-kernel")
            assert not torch.equal(out["1"]["y"], out["0"]["y"])


@pytest.mark.parametrize("halo", ["1", "0"], ids=["halo", "nt-glds"])
@pytest.mark.parametrize("planes", ["1", "0"], ids=["wgrad-planes", "wgrad-in-kernel"])
@pytest.mark.parametrize("case", [(128, 128, 3, "same", 32, 32, 2), (256, 256, 3, "down", 32, 32, 2), (64, 128, 4, "same", 31, 31, 2),
                                  (256, 256, 1, "same", 16, 16, 4), (8, 64, 3, "same", 64, 64, 2), (128, 8, 3, "same", 64, 64, 2),
                                  (4, 64, 3, "same", 64, 64, 2), (128, 4, 3, "same", 64, 64, 2)],
                         ids=lambda c: "-".join(map(str, c)))
def test_fp32_split_bf16_products(dev, case, planes, halo, monkeypatch):
    """`fp32x3` (dvq_set_fp32_split): fp32 tensors, every matrix product as three bf16 MFMA passes on two-plane operands.  Forward,
    input gradient, weight and bias gradient of a convolution against the exact-fp32 instantiation of the same kernels and against
    float64: the split products must be ~2^-17-accurate (two orders of magnitude inside north_star's 1e-3), not bf16-accurate"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    monkeypatch.setenv("DVQ_X3_WGRAD_PLANES", planes)     # weight gradient: bf16 planes + three launches of the bf16 kernels / in-kernel split
    monkeypatch.setenv("DVQ_X3_HALO", halo)               # 3 x 3 forward / input gradient: halo kernel on concatenated planes / in-kernel split
    cin, cout, k, kind, h, w_, n = case
    rs = np.random.RandomState(cin + 3 * cout + k)
    kw = dict(same=dict(stride=1, padding=(k - 1) // 2), down=dict(stride=2, padding=0, asym_pad=True))[kind]
    mod = Conv2d(cin, cout, k, **kw).to(dev)
    x = T(rs.standard_normal((n, cin, h, w_)).astype(np.float32), dev)
    res = {}
    for mode in ("fp32", "fp32x3"):
        with rt.compute_dtype_ctx(mode):
            assert rt.fp32_split() == (mode == "fp32x3")
            mod.weight.grad = mod.bias.grad = None
            xt = x.clone().requires_grad_(True)
            y = mod(xt)
            go = T(np.random.RandomState(5).standard_normal(tuple(y.shape)).astype(np.float32), dev)
            (y * go).sum().backward()
            res[mode] = [t.detach().double().cpu() for t in (y, xt.grad, mod.weight.grad, mod.bias.grad)]
    assert not rt.fp32_split()
    # float64 reference on the CPU
    xr = x.double().cpu().requires_grad_(True)
    wr, br = mod.weight.detach().double().cpu().requires_grad_(True), mod.bias.detach().double().cpu().requires_grad_(True)
    xin = torch.nn.functional.pad(xr, (0, 1, 0, 1)) if kind == "down" else xr
    yr = F.conv2d(xin, wr, br, stride=kw["stride"], padding=0 if kind == "down" else kw["padding"])
    (yr * go.double().cpu()).sum().backward()
    ref = [yr.detach(), xr.grad, wr.grad, br.grad]
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for name, a3, a1, r in zip(("y", "dx", "dw", "db"), res["fp32x3"], res["fp32"], ref):
        e3, e1 = rel(a3, r), rel(a1, r)
        assert e1 < 5e-6, (name, e1)
        assert e3 < 3e-5, (name, e3)                      # bf16 products would sit at ~3e-3
    assert rel(res["fp32x3"][0], res["fp32"][0]) > 0        # the split path really ran (it is not bit-identical to fp32 MFMA)


def test_fp32_split_gemms(dev):
    """plain NT / TN products (Linear layers, attention GEMMs) in fp32x3 against float64"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    torch.manual_seed(4)
    m, n, k = 777, 264, 520
    a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    ref = a.double() @ b.double().t()
    c, d = torch.randn(m, 136, device=dev), torch.randn(m, 200, device=dev)
    ref_tn = c.double().t() @ d.double()
    for mode, tol in (("fp32", 3e-6), ("fp32x3", 3e-5)):
        with rt.compute_dtype_ctx(mode):
            got = K.gemm_nt(a, b, m, n, k, k, k, n).view(m, n).double()
            got_tn = K.gemm_tn(c, d, m, 136, 200, 136, 200, 200).view(136, 200).double()
        assert float((got - ref).norm() / ref.norm()) < tol, mode
        assert float((got_tn - ref_tn).norm() / ref_tn.norm()) < tol, mode


DET_CASES = [(256, 256, 3, "down", 32, 32, 4), (256, 512, 4, "same", 16, 16, 4), (256, 256, 1, "same", 32, 32, 8), (128, 128, 3, "s1p1", 64, 64, 4),
             (128, 8, 3, "s1p1", 64, 64, 4), (8, 64, 3, "s1p1", 64, 64, 4), (512, 512, 3, "s1p1", 16, 16, 8)]


@pytest.mark.parametrize("case", DET_CASES, ids=lambda c: "-".join(map(str, c)))
def test_deterministic_weight_gradients(dev, case):
    """dvq_set_deterministic(1): the split-reduction weight-gradient families (patch-stage / transpose-read / halo incl. its thin variant /
    plain TN) give BIT-identical gradients launch after launch -- and the same values as the default (atomic) path up to summation order"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.layers import Conv2d
    cin, cout, k, kind, h, w_, n = case
    rs = np.random.RandomState(3 * cin + cout + k)
    kw = dict(same=dict(stride=1, padding=(k - 1) // 2), down=dict(stride=2, padding=0, asym_pad=True), s1p1=dict(stride=1, padding=1))[kind]
    mod = Conv2d(cin, cout, k, **kw).to(dev)
    x = T(bf16_round(rs.standard_normal((n, cin, h, w_)).astype(np.float32)), dev)
    grads = {}
    try:
        with rt.compute_dtype_ctx(torch.bfloat16):
            for det, reps in ((True, 4), (False, 1)):
                K.set_deterministic(det)
                assert K.deterministic() == det
                out = []
                for _ in range(reps):
                    mod.weight.grad = mod.bias.grad = None
                    xt = x.clone().requires_grad_(True)
                    y = mod(xt)
                    go = T(bf16_round(np.random.RandomState(5).standard_normal(tuple(y.shape)).astype(np.float32)), dev)
                    (y.float() * go).sum().backward()
                    out.append((mod.weight.grad.clone(), mod.bias.grad.clone()))
                grads[det] = out
    finally:
        K.set_deterministic(False)
    w0, b0 = grads[True][0]
    for w_i, b_i in grads[True][1:]:
        assert torch.equal(w_i, w0) and torch.equal(b_i, b0), "deterministic mode: gradients differ between identical launches"
    wa, ba = grads[False][0]
    assert float((wa - w0).norm()) <= 1e-5 * float(w0.norm()) and float((ba - b0).norm()) <= 1e-5 * float(b0.norm())


def test_deterministic_linear_weight_gradient(dev):
    """the plain TN family (Linear weight gradients on gemm_tn_wide_pipe): partials + fold for every split count in deterministic mode"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.layers import Linear, Tape
    torch.manual_seed(2)
    try:
        with rt.compute_dtype_ctx(torch.bfloat16):
            for m, n_in, n_out in ((2048, 1024, 1024), (20736, 1024, 4096), (4096, 512, 264)):
                lin = Linear(n_in, n_out).to(dev)
                x = torch.randn(m, n_in, device=dev).to(torch.bfloat16)
                dy = torch.zeros(m, lin.out_p, device=dev, dtype=torch.bfloat16)
                dy[:, :n_out] = torch.randn(m, n_out, device=dev).to(torch.bfloat16)
                res = {}
                for det, reps in ((True, 3), (False, 1)):
                    K.set_deterministic(det)
                    out = []
                    for _ in range(reps):
                        lin.weight.grad = lin.bias.grad = None
                        tape = Tape()
                        lin.fwd(x, tape)
                        lin.bwd(dy, tape)
                        rt.join_side()
                        out.append((lin.weight.grad.clone(), lin.bias.grad.clone()))
                    res[det] = out
                w0, b0 = res[True][0]
                for w_i, b_i in res[True][1:]:
                    assert torch.equal(w_i, w0) and torch.equal(b_i, b0), (m, n_in, n_out)
                assert float((res[False][0][0] - w0).norm()) <= 1e-5 * float(w0.norm())
    finally:
        K.set_deterministic(False)


HALO_CASES = [(64, 64, 8, 32, 2), (128, 128, 16, 32, 2), (64, 192, 8, 64, 1), (256, 128, 24, 32, 1), (128, 64, 8, 32, 1),
              (128, 8, 16, 32, 1), (64, 24, 8, 32, 2), (64, 40, 8, 64, 1)]      # thin outputs: 32- / 64-wide channel tiles


@pytest.mark.parametrize("case", HALO_CASES, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("residual", [False, True])
def test_conv3x3_halo_kernel(dev, case, residual):
    """LDS-resident-halo kernel (impl 4) for 3x3/s1/p1 bf16: forward (+bias, +residual) and dgrad vs fp32 reference"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    cin, cout, h, w_, n = case
    rs = np.random.RandomState(cin + cout + h)
    x = bf16_round(rs.standard_normal((n, cin, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    res = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
    go = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
    xr = torch.from_numpy(x).requires_grad_(True)
    wr, br = torch.from_numpy(wt).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    if residual:
        yr = yr + torch.from_numpy(res)
    (yr * torch.from_numpy(go)).sum().backward()
    mod = Conv2d(cin, cout, 3, 1, 1).to(dev)
    with torch.no_grad():
        mod.weight.copy_(T(wt, dev))
        mod.bias.copy_(T(b, dev))
    with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(4):
        xh = T(x, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        rh = T(res, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous() if residual else None
        tape = Tape()
        y = mod.fwd(xh, tape, residual=rh)
        # the dgrad of a conv whose Cout is not a multiple of 64 is not a halo-kernel shape (its "input" channels): auto
        if cout % 64 != 0:
            tape.s["d"].impl = 0
        with rt.impl_ctx(4 if cout % 64 == 0 else 0):
            dx = mod.bwd(T(go, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous(), tape)
    yref = yr.detach().numpy()
    err = np.abs(y.float().permute(0, 3, 1, 2).cpu().numpy() - yref).max() / np.abs(yref).max()
    assert err < 2e-2, f"forward rel-to-max error {err}"
    dref = xr.grad.numpy()
    derr = np.abs(dx.float().permute(0, 3, 1, 2).cpu().numpy() - dref).max() / np.abs(dref).max()
    assert derr < 2e-2, f"dgrad rel-to-max error {derr}"
    for name, got, ref in (("dw", mod.weight.grad.cpu().numpy(), wr.grad.numpy()), ("db", mod.bias.grad.cpu().numpy(), br.grad.numpy())):
        e = np.abs(got - ref).max() / np.abs(ref).max()
        assert e < 2e-2, f"{name} rel-to-max error {e}"


@pytest.mark.parametrize("case", [(256, 256, 32, 32, 2), (512, 512, 16, 16, 3), (256, 512, 24, 20, 1), (320, 256, 16, 16, 2)],
                         ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("residual", [False, True])
def test_conv1x1_on_pipelined_gemm(dev, case, residual):
    """1x1 / stride 1 convolutions with >= 256 output channels (AttnBlock q / k / v / proj_out, nin_shortcut) run as plain GEMMs on
    the pipelined 256-wide kernel (automatic): forward (+bias, +residual), input gradient, weight / bias gradients against an
    fp32 torch reference and against the 128 x 128 implicit-GEMM kernel (impl 2); pixel counts that are no multiple of the tile"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    cin, cout, h, w_, n = case
    rs = np.random.RandomState(cin + cout + h)
    x = bf16_round(rs.standard_normal((n, cin, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    res = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
    go = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
    xr = torch.from_numpy(x).requires_grad_(True)
    wr, br = torch.from_numpy(wt).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    if residual:
        yr = yr + torch.from_numpy(res)
    (yr * torch.from_numpy(go)).sum().backward()
    outs = {}
    for impl in (0, 2):
        mod = Conv2d(cin, cout, 1).to(dev)
        with torch.no_grad():
            mod.weight.copy_(T(wt, dev))
            mod.bias.copy_(T(b, dev))
        with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(impl):
            xh = T(x, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous()
            rh = T(res, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous() if residual else None
            tape = Tape()
            y = mod.fwd(xh, tape, residual=rh)
            dx = mod.bwd(T(go, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous(), tape)
        outs[impl] = (y.float().permute(0, 3, 1, 2).cpu().numpy(), dx.float().permute(0, 3, 1, 2).cpu().numpy(),
                      mod.weight.grad.cpu().numpy(), mod.bias.grad.cpu().numpy())
    refs = (yr.detach().numpy(), xr.grad.numpy(), wr.grad.numpy(), br.grad.numpy())
    for impl, got in outs.items():
        for name, g, r in zip(("y", "dx", "dw", "db"), got, refs):
            e = np.abs(g - r).max() / np.abs(r).max()
            assert e < 2e-2, f"impl {impl} {name} rel-to-max error {e}"
    # same bf16 operands, same fp32 accumulation order up to tiling: the two kernels agree much closer than either with fp32
    assert np.abs(outs[0][0] - outs[2][0]).max() / np.abs(refs[0]).max() < 8e-3


@pytest.mark.parametrize("cout", [128, 64, 40])
def test_conv3x3_thin_input_kernel(dev, cout):
    """image heads on the thin-K kernel (bf16): 3 -> cout forward (+bias, ReLU) and the dgrad of a cout -> 3 conv"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    rs = np.random.RandomState(cout)
    n, h, w_ = 2, 12, 64
    x = bf16_round(rs.standard_normal((n, 3, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, 3, 3, 3)) / np.sqrt(27)).astype(np.float32))
    b = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    yr = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), padding=1)).numpy()
    with rt.compute_dtype_ctx(torch.bfloat16):
        mod = Conv2d(3, cout, 3, 1, 1).to(dev)
        with torch.no_grad():
            mod.weight.copy_(T(wt, dev))
            mod.bias.copy_(T(b, dev))
        xp = K.nchw_to_nhwc_pad(T(x, dev), 8, torch.bfloat16)
        y = mod.fwd(xp, None, act=K.ACT_RELU)
        err = np.abs(y.float().permute(0, 3, 1, 2).cpu().numpy()[:, :cout] - yr).max() / np.abs(yr).max()
        assert err < 1e-2, f"forward rel-to-max error {err}"
        # weight / bias gradient of the 3-channel input conv (taps-as-columns tile of the TN kernel)
        gy = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
        wr_, br_ = torch.from_numpy(wt).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
        (F.conv2d(torch.from_numpy(x), wr_, br_, padding=1) * torch.from_numpy(gy)).sum().backward()
        tp = Tape()
        mod.fwd(xp, tp)
        mod.bwd(T(gy, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous(), tp, need_dw=True)
        for name, got, ref in (("dw", mod.weight.grad, wr_.grad), ("db", mod.bias.grad, br_.grad)):
            e = np.abs(got.float().cpu().numpy() - ref.numpy()).max() / np.abs(ref.numpy()).max()
            assert e < 1e-2, f"{name} rel-to-max error {e}"
        # dgrad of cout -> 3: gradient [n,h,w,8 (3 real)] -> [n,h,w,cout]
        mod2 = Conv2d(cout, 3, 3, 1, 1).to(dev)
        w2 = bf16_round((rs.standard_normal((3, cout, 3, 3)) / np.sqrt(cout * 9)).astype(np.float32))
        with torch.no_grad():
            mod2.weight.copy_(T(w2, dev))
        xin = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
        go = bf16_round(rs.standard_normal((n, 3, h, w_)).astype(np.float32))
        xr = torch.from_numpy(xin).requires_grad_(True)
        (F.conv2d(xr, torch.from_numpy(w2), None, padding=1) * torch.from_numpy(go)).sum().backward()
        tape = Tape()
        xh = T(xin, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        mod2.fwd(xh, tape)
        dx = mod2.bwd(K.nchw_to_nhwc_pad(T(go, dev), 8, torch.bfloat16), tape, need_dw=False)
        dref = xr.grad.numpy()
        derr = np.abs(dx.float().permute(0, 3, 1, 2).cpu().numpy() - dref).max() / np.abs(dref).max()
        assert derr < 1e-2, f"dgrad rel-to-max error {derr}"


@pytest.mark.parametrize("shape", [(20736, 1024), (4097, 136), (100, 8), (63, 64), (3000, 12), (5, 4096)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sum_batch(dev, shape, dtype):
    """bias-gradient column sums (accumulating): the 16-byte kernel (inner % 8 == 0, batch >= 64) and the scalar one"""
    from dynamicvectorquantization_amd import kernels as K
    rows, inner = shape
    rs = np.random.RandomState(rows + inner)
    x = rs.standard_normal((rows, inner)).astype(np.float32)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    init = rs.standard_normal(inner).astype(np.float32)
    out = T(init, dev)
    K.sum_batch(T(x, dev, dtype), out)
    ref = init.astype(np.float64) + x.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.sqrt(rows))


@pytest.mark.parametrize("case", [(3, 3, 17, 23, 0), (2, 1, 8, 8, 0), (2, 4, 33, 5, 0), (5, 3, 64, 64, 16), (2, 6, 9, 7, 0)],
                         ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nchw_to_nhwc_pad_layout(dev, case, dtype):
    """image entry layout: the one-pixel-per-thread kernel (C <= 4 into one 16-byte channel vector) and the element-wise kernel
    (wider pads / more planes) -- exact copy with zero padding channels, and its inverse"""
    from dynamicvectorquantization_amd import kernels as K
    b, c, h, w, cp = case
    vec = 4 if dtype == torch.float32 else 8
    cp = cp or vec * -(-c // vec)
    x = np.random.RandomState(b * 100 + c).standard_normal((b, c, h, w)).astype(np.float32)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    out = K.nchw_to_nhwc_pad(T(x, dev), cp, dtype)
    ref = np.zeros((b, h, w, cp), np.float32)
    ref[..., :c] = x.transpose(0, 2, 3, 1)
    assert np.array_equal(out.float().cpu().numpy(), ref)
    assert np.array_equal(K.nhwc_pad_to_nchw(out, c).cpu().numpy(), x)


@pytest.mark.parametrize("case", [(64, 2, 16, 24), (64, 1, 64, 64), (32, 3, 6, 10)], ids=lambda c: "-".join(map(str, c)))
def test_tconv4x4s2_thin_kernel(dev, case):
    """input gradient of the PatchGAN's first conv (4x4 / s2 / p1, 3 image channels) on the thin transposed-conv kernel"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    cout, n, h, w_ = case
    rs = np.random.RandomState(cout + h)
    x = bf16_round(rs.standard_normal((n, 3, h, w_)).astype(np.float32))
    wt = bf16_round((rs.standard_normal((cout, 3, 4, 4)) / np.sqrt(48)).astype(np.float32))
    go = bf16_round(rs.standard_normal((n, cout, h // 2, w_ // 2)).astype(np.float32))
    xr = torch.from_numpy(x).requires_grad_(True)
    (F.conv2d(xr, torch.from_numpy(wt), None, stride=2, padding=1) * torch.from_numpy(go)).sum().backward()
    with rt.compute_dtype_ctx(torch.bfloat16):
        mod = Conv2d(3, cout, 4, 2, 1).to(dev)
        with torch.no_grad():
            mod.weight.copy_(T(wt, dev))
        tape = Tape()
        mod.fwd(K.nchw_to_nhwc_pad(T(x, dev), 8, torch.bfloat16), tape)
        outs, dws = [], []
        for impl in (0, 3):          # thin kernels (auto) and the generic implicit-GEMM dgrad / per-tap weight gradient
            tape.s["d"].impl = impl
            mod.weight.grad = None
            dx = mod.bwd(T(go, dev, torch.bfloat16).permute(0, 2, 3, 1).contiguous(), tape, need_dw=True)
            outs.append(dx.float().permute(0, 3, 1, 2).cpu().numpy())
            dws.append(mod.weight.grad.float().cpu().numpy().copy())
    wr = torch.from_numpy(wt).requires_grad_(True)
    (F.conv2d(torch.from_numpy(x), wr, None, stride=2, padding=1) * torch.from_numpy(go)).sum().backward()
    for dw in dws:              # weight gradient of the 3-channel input conv: taps-as-columns tile of the TN kernel
        assert np.abs(dw - wr.grad.numpy()).max() / np.abs(wr.grad.numpy()).max() < 1e-2
    dref = xr.grad.numpy()
    for got in outs:
        assert np.all(got[:, 3:] == 0), "pad channels must stay zero"
        derr = np.abs(got[:, :3] - dref).max() / np.abs(dref).max()
        assert derr < 1e-2, f"dgrad rel-to-max error {derr}"
    assert np.abs(outs[0] - outs[1]).max() / np.abs(dref).max() < 1e-2


@pytest.mark.parametrize("impl", [5, 6, 7, 8, 10], ids=["plain", "pipelined", "pipelined-4wave", "pipelined-192", "8phase"])
@pytest.mark.parametrize("shape", [(3, 600, 520, 200, 1), (1, 2048, 1032, 512, 2), (2, 1296, 648, 128, 0), (1, 300, 264, 64, 1),
                                   (1, 1100, 776, 448, 1), (2, 520, 512, 4160, 2)])
def test_gemm_nt_wide_kernel(dev, shape, impl):
    """256 x 256 macro-tile GEMMs (bf16; impl 5 = plain main loop, impl 6 = the software-pipelined one the automatic dispatch uses
    for large plain products the library does not take): ragged M / N / K tails, a single K slab, bias per column / row, batches --
    against an fp32 torch product of the same bf16-rounded operands"""
    from dynamicvectorquantization_amd import kernels as K
    b, m, n, k, bias_mode = shape
    rs = np.random.RandomState(m + n)
    a = bf16_round(rs.standard_normal((b, m, k)).astype(np.float32))
    w = bf16_round(rs.standard_normal((b, n, k)).astype(np.float32) / np.sqrt(k))
    bias = rs.standard_normal(n if bias_mode == 1 else m).astype(np.float32) if bias_mode else None
    ref = torch.from_numpy(a) @ torch.from_numpy(w).transpose(1, 2)
    if bias_mode == 1:
        ref = ref + torch.from_numpy(bias)[None, None, :]
    elif bias_mode == 2:
        ref = ref + torch.from_numpy(bias)[None, :, None]
    at, wt_ = T(a, dev, torch.bfloat16).reshape(-1), T(w, dev, torch.bfloat16).reshape(-1)
    bt = T(bias, dev) if bias_mode else None
    out = K.gemm_nt(at, wt_, m, n, k, k, k, n, batch=b, sa=m * k, sb=n * k, sc=m * n, bias=bt, bias_mode=bias_mode, impl=impl)
    got = out.view(b, m, n).float().cpu()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-2, err


def test_gemm_nt_8phase_long_reduction(dev):
    """the 8-phase 256 x 256 kernel (counted vmcnt, DMA queue never drained) where the automatic dispatch takes it -- K >= 4096 over at
    least two rounds of tiles -- and forced (impl 10) on an odd number of K tiles with ragged row / column tails: the WHOLE output against
    an fp32 product of the same bf16 operands, and bit-identical results over repeated launches"""
    from dynamicvectorquantization_amd import kernels as K
    torch.manual_seed(11)
    for (m, n, k, impl) in ((8192, 4096, 4096, 0), (4100, 2056, 4160, 10), (8192, 8192, 8192, 0)):
        a2 = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
        b2 = (torch.rand(n, k, device=dev) * 2 - 1).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        outs = [K.gemm_nt(a2.reshape(-1), b2.reshape(-1), m, n, k, k, k, n, bias=bias, bias_mode=1, alpha=0.5, impl=impl) for _ in range(4)]
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        worst = 0.0
        for r0 in range(0, m, 2048):
            ref = 0.5 * torch.matmul(a2[r0:r0 + 2048].float(), b2.float().t()) + bias[None, :]
            worst = max(worst, float((outs[0].view(m, n)[r0:r0 + 2048].float() - ref).abs().max() / ref.abs().max()))
        assert worst < 6e-3, (m, n, k, worst)            # bf16 output rounding (2^-9 of the largest entry) + accumulation order


@pytest.mark.parametrize("shape", [(2048, 1024, 1024), (1304, 512, 1032), (20736, 4096, 1024)], ids=lambda s: "x".join(map(str, s)))
def test_linear_input_gradient(dev, shape):
    """Linear.bwd's input gradient dx = dy W: the automatic route (pipelined 256-wide NT kernel on a transposed weight copy) against
    the 128 x 128 kernel (DVQ impl 2) and an fp32 product"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Linear, Tape
    m, n_in, n_out = shape
    rs = np.random.RandomState(m + n_in)
    x = bf16_round(rs.standard_normal((m, n_in)).astype(np.float32))
    dy = bf16_round(rs.standard_normal((m, n_out)).astype(np.float32))
    with rt.compute_dtype_ctx(torch.bfloat16):
        lin = Linear(n_in, n_out).to(dev)
        with torch.no_grad():
            lin.weight.copy_(lin.weight.to(torch.bfloat16).float())
        wref = lin.weight.detach().float().cpu()
        ref = torch.from_numpy(dy)[:, :n_out] @ wref
        outs = []
        for impl in (0, 2):
            with rt.impl_ctx(impl):
                tape = Tape()
                lin.fwd(T(x, dev, torch.bfloat16), tape)
                dyp = torch.zeros(m, lin.out_p, dtype=torch.bfloat16, device=dev)
                dyp[:, :n_out] = T(dy, dev, torch.bfloat16)
                outs.append(lin.bwd(dyp, tape).float().cpu())
    for got in outs:
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-2
    assert float((outs[0] - outs[1]).abs().max()) / float(ref.abs().max()) < 1e-2


def test_linear_multi_pack(dev):
    """dvq_linear_pack_multi: ONE launch writes the bf16 copies w [out_p, in] and wt [in, out_p] of every Linear of an optimizer group --
    bit-equal to the per-layer cast + transpose (rows / columns past `out` zero), refreshed when the parameters change, padded heads
    and ragged widths included"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import LINEAR_PACKS, Linear
    torch.manual_seed(5)
    shapes = [(1024, 1024), (1024, 4096), (4096, 1024), (1024, 1026), (100, 72), (64, 8), (132, 515)]
    lins = [Linear(i, o, bias=(k % 2 == 0)).to(dev) for k, (i, o) in enumerate(shapes)]
    x = torch.randn(4, 8, device=dev)

    def check():
        for lin in lins:
            w, b = lin._w(torch.bfloat16)
            wt = lin._wt(w)
            ref = torch.zeros(lin.out_p, lin.in_features, device=dev)
            ref[: lin.out_features] = lin.weight.detach()
            ref = ref.to(torch.bfloat16)
            assert w.shape == ref.shape and torch.equal(w, ref), (lin.in_features, lin.out_features)
            assert wt.shape == (lin.in_features, lin.out_p) and torch.equal(wt, ref.t().contiguous())
            if lin.bias is not None:
                assert b.shape[0] == lin.out_p and torch.equal(b[: lin.out_features], lin.bias.detach())
                assert float(b[lin.out_features:].abs().sum()) == 0.0

    check()                                   # first use: each layer packs itself (table of one)
    ptrs = [lin._lpack["w"].data_ptr() for lin in lins]
    with torch.no_grad():
        for lin in lins:
            lin.weight.mul_(-1.5)              # _version changes: stale copies must be noticed ...
            if lin.bias is not None:
                lin.bias.add_(1.0)
    check()                                   # ... and now ONE launch refreshes all of them
    assert [lin._lpack["w"].data_ptr() for lin in lins] == ptrs          # persistent buffers
    assert LINEAR_PACKS.tables, "the group table was not built"
    rt.bump_weights_epoch()
    with torch.no_grad():
        lins[2].weight.copy_(torch.randn_like(lins[2].weight))
    check()
    # the layer's products run on the packed copies
    y = lins[4].fwd(torch.randn(256, 100, device=dev).to(torch.bfloat16), None)
    assert y.shape == (256, 72) and bool(torch.isfinite(y.float()).all())


@pytest.mark.parametrize("shape", [(8, 1024, 1024, 1), (1, 1032, 1024, 0), (32, 4096, 1024, 1), (8, 1024, 4096, 1), (13, 72, 200, 1), (32, 264, 64, 0)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_nt_skinny_kernel(dev, shape):
    """weight-streaming kernel of the single-token Linear layers (bf16, M <= 32, impl 0) against the 128 x 128 kernel (impl 2) and fp32"""
    from dynamicvectorquantization_amd import kernels as K
    m, n, k, with_bias = shape
    rs = np.random.RandomState(m * 7 + n + k)
    a = bf16_round(rs.standard_normal((m, k)).astype(np.float32))
    w = bf16_round(rs.standard_normal((n, k)).astype(np.float32) / np.sqrt(k))
    bias = rs.standard_normal(n).astype(np.float32) if with_bias else None
    ref = torch.from_numpy(a) @ torch.from_numpy(w).t()
    if with_bias:
        ref = ref + torch.from_numpy(bias)[None, :]
    at, wt_ = T(a, dev, torch.bfloat16).reshape(-1), T(w, dev, torch.bfloat16).reshape(-1)
    bt = T(bias, dev) if with_bias else None
    for impl in (0, 2):
        out = K.gemm_nt(at, wt_, m, n, k, k, k, n, bias=bt, bias_mode=1 if with_bias else 0, impl=impl)
        got = out.view(m, n).float().cpu()
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-2, impl


@pytest.mark.parametrize("shape", [(2048, 1024, 1024, 1), (1304, 1032, 512, 0), (20736, 1024, 4096, 1), (1024, 256, 264, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_nt_large_plain(dev, shape):
    """large plain bf16 products (impl 0, batch 1, bias per column or none) on the pipelined 256-wide kernel and on the 128 x 128
    kernel (impl 2): both against an fp32 product of the same bf16-rounded operands, and against each other"""
    from dynamicvectorquantization_amd import kernels as K
    m, n, k, with_bias = shape
    rs = np.random.RandomState(m + n + k)
    a = bf16_round(rs.standard_normal((m, k)).astype(np.float32))
    w = bf16_round(rs.standard_normal((n, k)).astype(np.float32) / np.sqrt(k))
    bias = rs.standard_normal(n).astype(np.float32) if with_bias else None
    ref = torch.from_numpy(a) @ torch.from_numpy(w).t()
    if with_bias:
        ref = ref + torch.from_numpy(bias)[None, :]
    at, wt_ = T(a, dev, torch.bfloat16).reshape(-1), T(w, dev, torch.bfloat16).reshape(-1)
    bt = T(bias, dev) if with_bias else None
    outs = []
    for impl in (0, 2):
        out = K.gemm_nt(at, wt_, m, n, k, k, k, n, bias=bt, bias_mode=1 if with_bias else 0, alpha=0.5 if not with_bias else 1.0, impl=impl)
        outs.append(out.view(m, n).float().cpu())
    if not with_bias:
        ref = ref * 0.5
    for got in outs:
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-2
    assert float((outs[0] - outs[1]).abs().max()) / float(ref.abs().max()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("impl", [1, 2])
def test_gemm_nt_tn(dev, dtype, impl):
    from dynamicvectorquantization_amd import kernels as K
    rs = np.random.RandomState(3)
    b, m, n, k = 3, 200, 72, 96
    a = rs.standard_normal((b, m, k)).astype(np.float32)
    bb = rs.standard_normal((b, n, k)).astype(np.float32)
    bias = rs.standard_normal(n).astype(np.float32)
    if dtype == torch.bfloat16:
        a, bb = bf16_round(a), bf16_round(bb)
    ref = 0.5 * np.einsum("bmk,bnk->bmn", a, bb) + bias
    out = K.gemm_nt(T(a, dev, dtype), T(bb, dev, dtype), m, n, k, k, k, n, batch=b, sa=m * k, sb=n * k, sc=m * n, alpha=0.5,
                    bias=T(bias, dev), bias_mode=1, impl=impl)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    err = np.abs(out.float().cpu().numpy().reshape(b, m, n) - ref).max() / np.abs(ref).max()
    assert err < tol, err
    # TN: C[i][j] = sum_m A[m][i] B[m][j]
    mred, i, j = 520, 72, 40
    a2 = rs.standard_normal((b, mred, i)).astype(np.float32)
    b2 = rs.standard_normal((b, mred, j)).astype(np.float32)
    if dtype == torch.bfloat16:
        a2, b2 = bf16_round(a2), bf16_round(b2)
    ref2 = np.einsum("bmi,bmj->bij", a2, b2)
    out2 = K.gemm_tn(T(a2, dev, dtype), T(b2, dev, dtype), mred, i, j, i, j, j, batch=b, sa=mred * i, sb=mred * j, sc=i * j,
                     impl=impl)
    err2 = np.abs(out2.cpu().numpy().reshape(b, i, j) - ref2).max() / np.abs(ref2).max()
    assert err2 < tol, err2


# ---------------------------------------------------------------------------------------------
# blocks against the reference goldens (fp32, forward + input/parameter gradients)
# ---------------------------------------------------------------------------------------------
def _load_block(mod, name, dev):
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(T(synth.det_param(name + "." + k, p.shape), dev))


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("name", ["res_32_64", "res_64_64", "attn_64", "attn_128", "down_64", "up_64"])
def test_blocks_golden(dev, name, impl):
    from dynamicvectorquantization_amd import layers as L
    from dynamicvectorquantization_amd import runtime as rt
    from test_oracle_golden import BLOCK_SHAPES
    g = load_golden("blocks")
    shapes, xshape, _ = BLOCK_SHAPES[name]
    ctor = {"res_32_64": lambda: L.ResnetBlock(in_channels=32, out_channels=64, temb_channels=0, dropout=0.0),
            "res_64_64": lambda: L.ResnetBlock(in_channels=64, out_channels=64, temb_channels=0, dropout=0.0),
            "attn_64": lambda: L.AttnBlock(64), "attn_128": lambda: L.AttnBlock(128),
            "down_64": lambda: L.Downsample(64, True), "up_64": lambda: L.Upsample(64, True)}[name]
    mod = ctor().to(dev)
    assert sorted(k for k, _ in mod.named_parameters()) == sorted(shapes)
    _load_block(mod, name, dev)
    x = T(synth.det_param(name + ".x", xshape) * 8.0, dev).requires_grad_(True)
    with rt.compute_dtype_ctx(torch.float32), rt.impl_ctx(impl):
        y = mod(x, None) if name.startswith("res") else mod(x)
        gout = T(synth.det_param(name + ".gout", tuple(y.shape)), dev)
        (y * gout).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[name + "_y"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[name + "_dx"], rtol=1e-3, atol=1e-3)
    for k, p in mod.named_parameters():
        ref = g[name + "_d." + k]
        # floor: gradients that are analytically zero (softmax is invariant to the k bias) are pure rounding noise
        s = max(1e-5, float(np.abs(ref).max()))
        assert float(np.abs(p.grad.cpu().numpy() - ref).max()) / s < 2e-3, k


@pytest.mark.parametrize("cin,cout", [(128, 128), (64, 128), (256, 256)])
def test_resnet_block_fused_groupnorm_bf16(dev, cin, cout):
    """bf16 ResnetBlock on a halo-eligible shape: the fused path (GroupNorm+swish applied to the conv's LDS input tile,
    GroupNorm statistics emitted by the producing conv's epilogue) against the unfused kernels and the fp32 oracle"""
    from dynamicvectorquantization_amd import layers as L
    from dynamicvectorquantization_amd import runtime as rt
    from oracle import dqvae as odq
    rs = np.random.RandomState(cin + cout)
    n, h, w_ = 2, 16, 32
    x = bf16_round((rs.standard_normal((n, cin, h, w_)) * 1.5 + 0.2).astype(np.float32))
    go = bf16_round(rs.standard_normal((n, cout, h, w_)).astype(np.float32))
    mod = L.ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=0, dropout=0.0).to(dev)
    with torch.no_grad():
        for k, p_ in mod.named_parameters():
            p_.copy_(T(synth.det_param("fused." + k, p_.shape), dev))
    res = {}
    for tag, impl in (("fused", 0), ("unfused", 2)):
        for p_ in mod.parameters():
            p_.grad = None
        rt.set_fuse_gn_prologue(tag == "fused")
        with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(impl):
            xt = T(x, dev).requires_grad_(True)
            y = mod(xt, None)
            (y.float() * T(go, dev)).sum().backward()
        res[tag] = dict(y=y.detach().float().cpu().numpy(), dx=xt.grad.float().cpu().numpy(),
                        **{k: p_.grad.cpu().numpy().copy() for k, p_ in mod.named_parameters()})
    rt.set_fuse_gn_prologue(False)
    sd = {"b." + k: torch.from_numpy(synth.det_param("fused." + k, p_.shape)).requires_grad_(True) for k, p_ in mod.named_parameters()}
    xr = torch.from_numpy(x).requires_grad_(True)
    yr = odq.resnet_block(sd, "b", xr)
    (yr * torch.from_numpy(go)).sum().backward()
    ref = dict(y=yr.detach().numpy(), dx=xr.grad.numpy(), **{k[2:]: v.grad.numpy() for k, v in sd.items()})
    for key in ref:
        s_ = max(1e-6, float(np.abs(ref[key]).max()))
        e_f = float(np.abs(res["fused"][key] - ref[key]).max()) / s_
        e_u = float(np.abs(res["unfused"][key] - ref[key]).max()) / s_
        assert e_f < 4e-2, f"fused {key}: rel-to-max error {e_f} (unfused {e_u})"
        assert e_u < 4e-2, f"unfused {key}: rel-to-max error {e_u}"


# ---- BASELINE-size property tests (size-independent identities, no CPU reference needed) -----------------------------
@pytest.mark.parametrize("shape", [(64, 256, 128, 128), (64, 64, 256, 256), (64, 32, 512, 512), (128, 256, 64, 64), (64, 256, 128, 3)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv3x3_adjoint_identities_full_size(dev, shape):
    """<conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)> (+ bias term) at the BASELINE batch / resolution: the three conv
    kernels of a layer are mutually consistent on the full-size launch geometry (tile tails, XCD remap, split-K)"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d, Tape
    n, h, cin, cout = shape
    torch.manual_seed(n + h + cin + cout)
    with rt.compute_dtype_ctx(torch.bfloat16):
        conv = Conv2d(cin, cout, 3, 1, 1).to(dev)
        with torch.no_grad():
            conv.weight.copy_(conv.weight.to(torch.bfloat16).float())       # bf16-representable parameters
            conv.bias.zero_()
        cin_p, cout_p = conv._padded(torch.bfloat16)
        x = torch.randn(n, h, h, cin_p, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, h, h, cout_p, device=dev).to(torch.bfloat16)
        if cout_p != cout:
            dy[..., cout:] = 0
        tape = Tape()
        y = conv.fwd(x, tape)
        conv.weight.grad = torch.zeros_like(conv.weight)
        conv.bias.grad = torch.zeros_like(conv.bias)
        dx = conv.bwd(dy, tape)
        a = float((y.double() * dy.double()).sum())
        b = float((x.double() * dx.double()).sum())
        c = float((conv.weight.detach().double() * conv.weight.grad.double()).sum())
        scale = float(y.double().pow(2).sum().sqrt() * dy.double().pow(2).sum().sqrt())
        # y and dx are rounded to bf16 once (relative 2^-9 per element, random sign): the inner products agree to ~1e-4 of
        # the Cauchy-Schwarz scale; wgrad accumulates in fp32 and is exact up to summation order
        assert abs(a - b) < 2e-4 * scale, (a, b, scale)
        assert abs(a - c) < 2e-4 * scale, (a, c, scale)
        # linearity in the input at full size: conv(2x) = 2 conv(x) exactly in bf16 (power-of-two scaling)
        # (repeated: an LDS race in the kernel -- e.g. a DMA still in flight when the epilogue re-uses the tile -- shows up as a
        # handful of wrong elements in one launch out of a few)
        y_twice = K.add(y, y)
        for _ in range(4):
            assert torch.equal(conv.fwd(K.add(x, x), None), y_twice)
        # the bias gradient is the plain sum of dy
        db = dy[..., :cout].double().sum(dim=(0, 1, 2))
        assert float((conv.bias.grad.double() - db).abs().max()) <= 1e-3 * float(db.abs().max() + 1)


def test_groupnorm_and_lpips_properties_full_size(dev):
    """GroupNorm output of the BASELINE-size tensor has zero mean / unit variance per (image, group); backward of a constant
    upstream gradient through a plain GroupNorm is zero; LPIPS(x, x) = 0 with zero gradient"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.losses import LPIPS
    torch.manual_seed(0)
    n, h, c = 64, 256, 128
    x = (torch.randn(n, h, h, c, device=dev) * 3 + 1.5).to(torch.bfloat16)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    y, mr = K.gn_forward(x, gamma, beta, 32, 1e-6, 0)
    yg = y.float().view(n, h * h, 32, c // 32)
    assert float(yg.mean(dim=(1, 3)).abs().max()) < 2e-3 and float((yg.var(dim=(1, 3), unbiased=False) - 1).abs().max()) < 5e-3
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    dx = K.gn_backward(x, torch.ones_like(x), mr, gamma, beta, dg, db, 32, 0)
    assert float(dx.float().abs().max()) < 2e-2                        # d/dx of sum(GroupNorm(x)) vanishes
    np.testing.assert_allclose(db.cpu().numpy(), np.full(c, n * h * h, dtype=np.float32), rtol=1e-6)
    with rt.compute_dtype_ctx(torch.bfloat16):
        lp = LPIPS().to(dev)
        img = K.nchw_to_nhwc_pad(torch.rand(8, 3, 256, 256, device=dev) * 2 - 1, 8, torch.bfloat16)
        val, d = lp.fwd(img, img.clone(), gscale=1.0)
        assert float(val.abs().max()) < 1e-12 and float(d.float().abs().max()) < 1e-9     # bf16 feature noise floor is ~1e-5


@pytest.mark.parametrize("shape", [(1, 2048, 512, 256), (1, 20736, 1024, 1032), (2, 1300, 264, 520), (1, 4096, 4096, 1024)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_tn_wide_pipelined_kernel(dev, shape):
    """256 x 256-tile TN GEMM of the Linear weight gradients (bf16, automatic for I, J >= 256, Mred >= 1024): ragged I / J tiles, a
    reduction length that is no multiple of the 64-row stage (rows past the end must read as zero), batches -- against the
    128 x 128 kernel (impl 2) and an fp32 product of the same bf16-rounded operands"""
    from dynamicvectorquantization_amd import kernels as K
    b, mred, i, j = shape
    rs = np.random.RandomState(mred + i)
    a = bf16_round(rs.standard_normal((b, mred, i)).astype(np.float32))
    w = bf16_round(rs.standard_normal((b, mred, j)).astype(np.float32))
    ref = torch.from_numpy(a).transpose(1, 2) @ torch.from_numpy(w)
    at, wt_ = T(a, dev, torch.bfloat16).reshape(-1), T(w, dev, torch.bfloat16).reshape(-1)
    outs = []
    for impl in (0, 2):
        out = K.gemm_tn(at, wt_, mred, i, j, i, j, j, batch=b, sa=mred * i, sb=mred * j, sc=i * j, impl=impl)
        outs.append(out.view(b, i, j).cpu())
    for got in outs:
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 2e-3


def test_pipelined_kernels_reproduce_bitwise_full_size(dev):
    """the software-pipelined kernels (halo conv forward with residual + statistics, its input gradient, the 256-wide NT GEMM in both
    tile heights, the TN GEMM with workspace fold) launched repeatedly on the same full-size operands reproduce their first result
    bit for bit: an LDS race (a DMA landing after its buffer was re-used, a missing barrier) shows as a few differing elements in
    one launch out of several (tools/debug/race_stress.py runs more shapes and repeats)"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    torch.manual_seed(3)
    K.ensure_workspace(dev)

    def same(fn, reps=10):
        ref = [t.clone() for t in fn()]
        for _ in range(reps):
            for a, b in zip(fn(), ref):
                assert torch.equal(a, b)

    with rt.compute_dtype_ctx(torch.bfloat16):
        conv = Conv2d(128, 128, 3, 1, 1).to(dev)
        w, wt, bias = conv.packed(torch.bfloat16)
        x = torch.randn(64, 256, 256, 128, device=dev).to(torch.bfloat16)
        r = torch.randn_like(x)
        d = conv._desc(x)

        def fwd():
            st = torch.zeros(64, 32, 2, dtype=torch.float64, device=dev)
            return [K.conv2d_fwd(d, x, w, bias, r, out_stats=st, out_groups=32), st]
        same(fwd)
        same(lambda: [K.conv2d_dgrad(d, r, wt)])
        del x, r
        m, n, k = 20736, 1024, 4096
        a = torch.randn(m, k, device=dev).to(torch.bfloat16).reshape(-1)
        b = torch.randn(n, k, device=dev).to(torch.bfloat16).reshape(-1)
        for impl in (6, 8, 10):
            same(lambda: [K.gemm_nt(a, b, m, n, k, k, k, n, impl=impl)])
        a2 = torch.randn(m, 1024, device=dev).to(torch.bfloat16).reshape(-1)
        b2 = torch.randn(m, 4096, device=dev).to(torch.bfloat16).reshape(-1)
        same(lambda: [K.gemm_tn(a2, b2, m, 1024, 4096, 1024, 4096, 4096)])
