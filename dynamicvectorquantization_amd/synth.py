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
