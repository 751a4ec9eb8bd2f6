"""Entropy-threshold calibration and evaluation inputs (n3, parts of n1 / n4).

  * threshold_table      the percentile rule of /root/reference/scripts/tools/calculate_entropy_thresholds.py:99-110
                         (sorted entropies, entry "i" = sorted[(size * i) // 100], i = 1..99) -- the JSON that
                         DualGrainFixedEntropyRouter reads (RouterDual.py:49-51)
  * patch_entropies      all patch entropies of an image set on the fused HIP kernel (dvq_patch_entropy_gate[_range]).
                         bins="model" uses the bins the MODEL evaluates at run time, linspace(-1, 1, 32)
                         (dqvae_dual_entropy.py:61); bins="reference" reproduces the reference script's linspace(0, 1, 32)
                         (:74) -- the two differ, which is why a table calibrated with the script does not give the nominal
                         fine ratio when the model applies it
  * load_images          evaluation preprocessing of data/imagenet_base.py:24-30 (Resize(256) on the shorter side, CenterCrop,
                         [0,1] -> [-1,1]) with PIL, or a ready .npy of [N,3,H,W] fp32 in [-1,1]
"""
from __future__ import annotations

import json
import os

import numpy as np

BINS = {"model": (-1.0, 1.0), "reference": (0.0, 1.0)}


def threshold_table(entropies) -> dict:
    e = np.sort(np.asarray(entropies, dtype=np.float32).reshape(-1))
    size = int(e.shape[0])
    if size < 100:
        raise ValueError(f"need at least 100 patch entropies for a 99-entry percentile table, got {size}")
    return {str(i + 1): float(e[(size * (i + 1)) // 100]) for i in range(99)}


def write_table(path: str, table: dict) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(table, f)


def default_table_path(dataset_type: str, split: str, patch_size: int, root: str = ".") -> str:
    """file name the reference script writes and the shipped YAMLs point at"""
    return os.path.join(root, "scripts/tools/thresholds", f"entropy_thresholds_{dataset_type}_{split}_patch-{patch_size}.json")


def patch_entropies(images: np.ndarray, patch: int = 16, bins="model", batch_size: int = 64, device="cuda") -> np.ndarray:
    """images [N,3,H,W] fp32 in [-1,1] -> flat fp32 array of N * (H/p) * (W/p) entropies (HIP kernel; no CPU fallback)"""
    import torch

    from . import kernels as K
    rng = BINS[bins] if isinstance(bins, str) else tuple(float(b) for b in bins)
    out = []
    for i in range(0, images.shape[0], batch_size):
        x = torch.from_numpy(np.ascontiguousarray(images[i:i + batch_size], dtype=np.float32)).to(device)
        ent, _ = K.patch_entropy_gate(x, patch, None, bins=rng)
        out.append(ent.reshape(-1).cpu().numpy())
    return np.concatenate(out)


def _eval_transform(img, size: int) -> np.ndarray:
    from PIL import Image
    if img.mode != "RGB":
        img = img.convert("RGB")
    from .data import resized_size
    w, h = img.size
    nw, nh = resized_size(w, h, size)                     # transforms.Resize(size): shorter side -> size, longer int(size * long / short)
    img = img.resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))        # transforms.CenterCrop(size)
    img = img.crop((left, top, left + size, top + size))
    a = np.asarray(img, dtype=np.float32) / 255.0          # ToTensor
    return ((a - 0.5) / 0.5).transpose(2, 0, 1)            # Normalize(0.5, 0.5)


def load_images(path: str, size: int = 256, limit: int | None = None) -> np.ndarray:
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
        assert a.ndim == 4 and a.shape[1] == 3, "expected [N,3,H,W]"
        return a[:limit] if limit else a
    from PIL import Image
    exts = (".png", ".jpg", ".jpeg", ".bmp", ".webp")
    files = sorted(os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.lower().endswith(exts))
    if limit:
        files = files[:limit]
    if not files:
        raise FileNotFoundError(f"no images under {path}")
    return np.stack([_eval_transform(Image.open(f), size) for f in files]).astype(np.float32)


def sequence_length_stats(grain_indices) -> dict:
    """scripts/tools/visualize_dual_grain.py:46-56: tokens per image = 1 per coarse cell + 4 per fine cell"""
    g = np.asarray(grain_indices)
    seq = (1 * (g == 0) + 4 * (g == 1)).reshape(g.shape[0], -1).sum(axis=1)
    return {"mean": float(np.mean(seq)), "variance": float(np.var(seq)), "max": int(seq.max()), "min": int(seq.min())}
