"""Deterministic synthetic inputs and parameters.

Used by bench.py, the tests and tools/gen_golden.py so that fixtures only have to
carry *outputs*: every input tensor and every model parameter can be regenerated
bit-for-bit from (name, shape, seed) with numpy's frozen legacy ``RandomState``
stream on both sides (the container that can import the reference, and the GPU
box that cannot).

Image recipe follows SURVEY.md section 8(d): the "half-flat" pattern gives a fine
ratio of exactly 0.5 under the shipped ImageNet threshold table (flat patches have
entropy well below 1.678, uniform-noise patches well above).
"""
from __future__ import annotations

import zlib

import numpy as np


def _rs(name: str, seed: int = 0) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 32))


def det_param(name: str, shape, seed: int = 0) -> np.ndarray:
    """Deterministic fp32 parameter for a state_dict entry.

    ndim >= 2 : U(-b, b), b = 1/sqrt(fan_in)         (conv / linear / embedding tables)
    ndim == 1 : '*weight' -> 1 + 0.1 U(-1,1) (norm gains); otherwise 0.1 U(-1,1) (biases)
    """
    shape = tuple(int(s) for s in shape)
    rs = _rs(name, seed)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return rs.uniform(-b, b, size=shape).astype(np.float32)
    if name.endswith("weight"):
        return (1.0 + 0.1 * rs.uniform(-1, 1, size=shape)).astype(np.float32)
    return (0.1 * rs.uniform(-1, 1, size=shape)).astype(np.float32)


def det_lpips_param(name: str, shape, seed: int = 0) -> np.ndarray:
    """Deterministic LPIPS parameters (ImageNet VGG16 weights cannot be fetched offline): conv weights are scaled by
    sqrt(6) so the 13-layer ReLU stack keeps its activation variance, NetLinLayer weights are non-negative."""
    v = det_param(name, shape, seed)
    if ".model." in name:
        return np.abs(v)
    if len(tuple(shape)) == 4:
        return (v * np.float32(np.sqrt(6.0))).astype(np.float32)
    return v


def det_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {name: shape}. Returns {name: np.float32 array}."""
    return {k: det_param(k, v, seed) for k, v in shapes.items()}


def half_flat_images(batch: int, size: int = 256, patch: int = 16, seed: int = 1234,
                     fine_fraction: float = 0.5) -> np.ndarray:
    """[B,3,size,size] fp32 in [-1,1]; per image a seeded random ``fine_fraction`` of the
    patch grid is U(-1,1) noise (high entropy -> fine grain), the rest is
    ``const_c + 0.05 N(0,1)`` with const_c ~ U(-0.8,0.8) per channel (low entropy -> coarse).
    """
    g = size // patch
    n_patch = g * g
    n_fine = int(round(n_patch * fine_fraction))
    out = np.empty((batch, 3, size, size), dtype=np.float32)
    for b in range(batch):
        rs = np.random.RandomState((seed * 1000003 + b) % (2 ** 32))
        perm = rs.permutation(n_patch)
        fine = np.zeros(n_patch, dtype=bool)
        fine[perm[:n_fine]] = True
        noise = rs.uniform(-1, 1, size=(3, size, size)).astype(np.float32)
        consts = rs.uniform(-0.8, 0.8, size=(n_patch, 3)).astype(np.float32)
        jitter = (0.05 * rs.standard_normal(size=(3, size, size))).astype(np.float32)
        img = np.empty((3, size, size), dtype=np.float32)
        for p in range(n_patch):
            i, j = divmod(p, g)
            sl = (slice(None), slice(i * patch, (i + 1) * patch), slice(j * patch, (j + 1) * patch))
            if fine[p]:
                img[sl] = noise[sl]
            else:
                img[sl] = consts[p][:, None, None] + jitter[sl]
        out[b] = np.clip(img, -1.0, 1.0)
    return out


def ragged_grain_images(size: int = 64, seed: int = 31, fractions=(0.5, 0.25, 0.875)) -> np.ndarray:
    """one image per fine-patch fraction: their coarse / fine code streams have different lengths (ragged stage-2 batches)"""
    return np.concatenate([half_flat_images(1, size, seed=seed + i, fine_fraction=f) for i, f in enumerate(fractions)], 0)


def vq_inputs(n: int, dim: int, k: int, dist: str = "normal", seed: int = 0):
    """VQ micro-benchmark inputs: x [n,dim], codebook [k,dim], fp32.

    dist = 'normal'  : x ~ N(0,1), codebook ~ N(0,1)
    dist = 'encoder' : x ~ N(0, 12/dim) (||z||^2 ~ 12), codebook ~ U(+-1/k) (reference init)
    """
    rs = np.random.RandomState((seed * 9176 + n + 31 * k) % (2 ** 32))
    if dist == "normal":
        x = rs.standard_normal((n, dim)).astype(np.float32)
        cb = rs.standard_normal((k, dim)).astype(np.float32)
    elif dist == "encoder":
        x = (rs.standard_normal((n, dim)) * np.sqrt(12.0 / dim)).astype(np.float32)
        cb = rs.uniform(-1.0 / k, 1.0 / k, size=(k, dim)).astype(np.float32)
    else:
        raise ValueError(dist)
    return x, cb


def dqvae_golden_state(keys, shapes, variant: str, k: int, zc: int) -> dict:
    """state_dict of the DQ-VAE goldens (tests/golden/dqvae_*.npz hold the key / shape lists and OUTPUTS only): every entry is
    `det_param(key, shape)`; the codebook is either `spread` (wide: score gaps far above rounding, so the reference's code indices
    are reproducible) or `refinit` (the reference's own U(+-1/K) initialisation: near ties).  numpy arrays, fp32."""
    sd = {}
    for kk, sh in zip(keys, shapes):
        sd[kk] = det_param(kk, sh) if len(sh) else np.zeros((), dtype=np.float32)
    if variant == "spread":
        cbw = det_param("quantize.codebook.weight.spread", (k + 1, zc)) * np.sqrt(zc) * 1.2
    else:
        cbw = np.random.RandomState(3).uniform(-1.0 / k, 1.0 / k, size=(k + 1, zc)).astype(np.float32)
    sd["quantize.codebook.weight"] = cbw.astype(np.float32)
    return sd


# geometries of the DQ-VAE fixtures: `small` = shrunken widths, `c1` = BASELINE config 1 (full width, 64 x 64 images)
DQVAE_GEOM = {"small": dict(ch=32, resolution=64, latent=8, zc=64, k=512, attn_enc=[4, 8], attn_dec=[8]),
              "c1": dict(ch=128, resolution=64, latent=8, zc=256, k=1024, attn_enc=[4, 8], attn_dec=[8])}


# ---- the pinned training step (tests/golden/train_step.npz; tools/gen_golden.py::gen_train_step) ------------------------------
def train_step_param(key: str, shape, k: int, zc: int) -> np.ndarray:
    """deterministic value of ONE parameter of the complete stage-1 model (autoencoder + `loss.discriminator.*` +
    `loss.perceptual_loss.*`), by state_dict key: the same naming the lossnet / dqvae fixtures use"""
    if key == "quantize.codebook.weight":
        return (det_param("quantize.codebook.weight.spread", (k + 1, zc)) * np.sqrt(zc) * 1.2).astype(np.float32)
    if key.startswith("loss.discriminator."):
        return det_param("disc." + key[len("loss.discriminator."):], shape)
    if key.startswith("loss.perceptual_loss."):
        return det_lpips_param("lpips." + key[len("loss.perceptual_loss."):], shape)
    return det_param(key, shape)


def train_step_vq_state(k: int, zc: int):
    """(cluster_size_ema, embed_ema) at the start of the pinned training run: even codes are LIVE (EMA count 3..7: they
    survive the decay and take the normalised-EMA path), odd codes start at the constructor's zeros (count < 1 after the first
    update: they take the restart path, quantize2_mask.py:102-105)"""
    cb = train_step_param("quantize.codebook.weight", (k + 1, zc), k, zc)[:-1]
    n0 = (5.0 + 2.0 * _rs("train_step.cluster_size_ema").uniform(-1, 1, size=(k,))).astype(np.float32)
    n0[1::2] = 0.0
    return n0, (cb * n0[:, None]).astype(np.float32)


def half_flat_layout(batch: int, size: int = 256, patch: int = 16, seed: int = 1234, fine_fraction: float = 0.5) -> np.ndarray:
    """bool [B, size/patch, size/patch]: which patches of half_flat_images(...) are the noise (fine-grain) ones"""
    g = size // patch
    n_fine = int(round(g * g * fine_fraction))
    out = np.zeros((batch, g * g), dtype=bool)
    for b in range(batch):
        rs = np.random.RandomState((seed * 1000003 + b) % (2 ** 32))
        out[b, rs.permutation(g * g)[:n_fine]] = True
    return out.reshape(batch, g, g)


def train_step_batches(steps: int, bs: int, size: int = 64):
    return [half_flat_images(bs, size, seed=7100 + s) for s in range(steps)]


def train_step_restart_perm(step: int, bs: int, k: int, size: int = 64) -> np.ndarray:
    """the permutation injected for torch.randperm(N) in the EMA updates of step `step` of the pinned run (quantize2_mask.py:97;
    N = bs * (size/8)^2 rows of the fine grid).  A coarse cell contributes FOUR identical rows (its feature vector is repeated 2x2), so
    a generic permutation restarts several codes with the same vector; the reference then resolves the resulting exact ties by the
    rounding noise of its fp32 addmm, which nothing can reproduce.  This permutation's first K entries are pairwise DISTINCT rows
    (every fine-grain row, one row per coarse cell) in seeded random order; the rest follow.  Any permutation is a legal draw."""
    fine = half_flat_layout(bs, size, 16, seed=7100 + step)              # [B, g, g] over 16-pixel patches = coarse cells
    g = fine.shape[1]
    hw = 2 * g
    distinct = []
    for b in range(bs):
        for i in range(hw):
            for j in range(hw):
                if fine[b, i // 2, j // 2] or (i % 2 == 0 and j % 2 == 0):
                    distinct.append(b * hw * hw + i * hw + j)
    distinct = np.array(distinct, dtype=np.int64)
    assert len(distinct) >= k, (len(distinct), k)
    rs = _rs(f"train_step.restart_perm.{step}")
    head = distinct[rs.permutation(len(distinct))]
    rest = np.setdiff1d(np.arange(bs * hw * hw, dtype=np.int64), head)
    return np.concatenate([head, rest[rs.permutation(len(rest))]])


def apply_train_step_state(model, k: int, zc: int, scale=None):
    """write the pinned run's start state into a stage-1 model (the reference's class or this repo's -- same parameter names):
    every parameter by name (times scale[name] where given), the VQ EMA buffers; BatchNorm / ScalingLayer buffers keep their
    constructor values"""
    import torch
    with torch.no_grad():
        for name, p in model.named_parameters():
            v = train_step_param(name, tuple(p.shape), k, zc)
            if scale and name in scale:
                v = (v * np.float32(scale[name])).astype(np.float32)
            p.copy_(torch.from_numpy(v).to(p.device))
        n0, s0 = train_step_vq_state(k, zc)
        cbm = model.quantize.codebook
        cbm.cluster_size_ema.copy_(torch.from_numpy(n0).to(cbm.cluster_size_ema.device))
        cbm.embed_ema.copy_(torch.from_numpy(s0).to(cbm.embed_ema.device))


def train_step_gumbel(step: int, bs: int, hc: int = 2, heads: int = 3) -> np.ndarray:
    """Exp(1) noise injected into F.gumbel_softmax's Tensor.exponential_ in BOTH forwards of step `step` of the pinned triple-grain run"""
    return np.random.RandomState(4211 + step).exponential(size=(bs, hc, hc, heads)).astype(np.float32)


def distinct_row_perm(indices: np.ndarray, n_heads: int, k: int, name: str) -> np.ndarray:
    """restart permutation whose first K entries are pairwise distinct rows, from a grain map: indices [B, hc, hc] in 0 .. n_heads-1
    (0 = coarsest) over the coarsest grid; a cell of grain g is repeated 2^(n_heads-1-g) times along both axes of the fine grid"""
    b, hc, _ = indices.shape
    f = 1 << (n_heads - 1)
    hw = hc * f
    distinct = []
    for bi in range(b):
        for i in range(hw):
            for j in range(hw):
                r = 1 << (n_heads - 1 - int(indices[bi, i // f, j // f]))
                if i % r == 0 and j % r == 0:
                    distinct.append(bi * hw * hw + i * hw + j)
    distinct = np.array(distinct, dtype=np.int64)
    rs = _rs(name)
    head = distinct[rs.permutation(len(distinct))]
    rest = np.setdiff1d(np.arange(b * hw * hw, dtype=np.int64), head)
    perm = np.concatenate([head, rest[rs.permutation(len(rest))]])
    assert len(distinct) >= k, (len(distinct), k)
    return perm
