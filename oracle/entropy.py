"""Oracle: per-patch soft-histogram entropy and the fixed-entropy router gate.

Follows (behaviour, not code):
  * Entropy.forward / .entropy   /root/reference/models/stage1_dynamic/dqvae_dual_entropy.py:25-63
  * DualGrainFixedEntropyRouter   /root/reference/modules/dynamic_modules/RouterDual.py:46-57

Math: gray = .2989 R + .5870 G + .1140 B; non-overlapping p x p patches; 32 bins on
linspace(-1,1,32); k(v,b) = exp(-.5 ((v-b)/sigma)^2), sigma = 0.01; pdf_b = mean_v k;
pdf = pdf / (sum pdf + 1e-40) + 1e-40 (1e-40 is an fp32 subnormal and must not be flushed);
H = - sum_b pdf log pdf.  Gate = [H <= t, H > t] with t = table[str(int(100 - r*100))].
"""
from __future__ import annotations

import json

import numpy as np
import torch


def patch_entropy(x: np.ndarray, patch: int = 16, nbins: int = 32, sigma: float = 0.01, bins=(-1.0, 1.0)) -> np.ndarray:
    """x [B,3,H,W] fp32 -> entropy [B,H/p,W/p] fp32 (torch-CPU fp32 arithmetic, denormals kept).  bins: range of the histogram
    (the model: (-1, 1); the reference's calibration script scripts/tools/calculate_entropy_thresholds.py:74: (0, 1))."""
    xt = torch.as_tensor(np.asarray(x, dtype=np.float32))
    b, _, h, w = xt.shape
    gh, gw = h // patch, w // patch
    gray = 0.2989 * xt[:, 0] + 0.5870 * xt[:, 1] + 0.1140 * xt[:, 2]            # [B,H,W]
    v = gray.reshape(b, gh, patch, gw, patch).permute(0, 1, 3, 2, 4).reshape(b * gh * gw, patch * patch)
    bins = torch.linspace(float(bins[0]), float(bins[1]), nbins)
    eps = 1e-40
    out = torch.empty(v.shape[0], dtype=torch.float32)
    step = 4096
    for s in range(0, v.shape[0], step):
        r = v[s:s + step].unsqueeze(2) - bins.view(1, 1, -1)
        k = torch.exp(-0.5 * (r / torch.tensor(sigma)).pow(2))
        pdf = k.mean(dim=1)
        pdf = pdf / (pdf.sum(dim=1, keepdim=True) + eps) + eps
        out[s:s + step] = -(pdf * torch.log(pdf)).sum(dim=1)
    return out.reshape(b, gh, gw).numpy()


def threshold_from_table(json_path: str, fine_grain_ratio: float) -> float:
    """Key arithmetic reproduced as written in RouterDual.py:51 -- int(100 - r*100) in Python
    floats with truncation (0.55 -> '44', 0.56 -> '43')."""
    with open(json_path, "r", encoding="utf-8") as f:
        table = json.load(f)
    return table[str(int(100 - fine_grain_ratio * 100))]


def entropy_gate(entropy: np.ndarray, threshold: float) -> np.ndarray:
    """[B,h,w] fp32 -> gate int64 [B,h,w,2] = [coarse, fine]; compare is fp32 vs the Python
    float threshold exactly like torch (the scalar is converted to the tensor dtype)."""
    t = np.float32(threshold)
    fine = (entropy > t)
    return np.stack([~fine, fine], axis=-1).astype(np.int64)


def threshold_table(entropies: np.ndarray) -> dict:
    """The percentile table of /root/reference/scripts/tools/calculate_entropy_thresholds.py:99-110: sort all patch
    entropies ascending; entry "i" (i = 1..99) = sorted[(size * i) // 100] as a Python float."""
    e = np.sort(np.asarray(entropies, dtype=np.float32).reshape(-1))
    size = e.shape[0]
    return {str(i + 1): float(e[int((size * (i + 1)) // 100)]) for i in range(99)}
