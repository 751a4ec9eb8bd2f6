"""Oracle: L2 codebook argmin, VectorQuantize2 forward, EMA codebook update.

Follows (behaviour, not code):
  * VQEmbedding.compute_distances / find_nearest_embedding
      /root/reference/modules/vector_quantization/quantize2_mask.py:29-55
  * VQEmbedding._update_buffers / _update_embedding   ... quantize2_mask.py:66-115
  * VectorQuantize2.forward                            ... quantize2_mask.py:157-191

The reference forms d = (|x|^2 + |e|^2) - 2 x.e in fp32 through a BLAS sgemm and takes the
first minimum.  Its result is only reproducible across BLAS back-ends where it coincides
with the mathematically exact argmin (SURVEY section 7 "Index-exact argmin"; BASELINE.md:
0 mismatches vs an fp64 recomputation).  The oracle therefore defines the answer as the
*exact* argmin of |x - e_k|^2 over the fp32 inputs, lowest index on ties, computed in fp64;
``argmin_fp32_formula`` restates the fp32 formula for comparison only.
"""
from __future__ import annotations

import numpy as np


def argmin_fp32_formula(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """fp32 restatement of the reference formula (quantize2_mask.py:39-46,53)."""
    x = np.asarray(x, dtype=np.float32)
    e = np.asarray(codebook, dtype=np.float32)
    xn = (x * x).sum(axis=1, keepdims=True, dtype=np.float32)
    en = (e * e).sum(axis=1, dtype=np.float32)[None, :]
    d = (xn + en) + np.float32(-2.0) * (x @ e.T)
    return d.argmin(axis=1).astype(np.int64)


def argmin_exact(x: np.ndarray, codebook: np.ndarray, return_gap: bool = False, chunk: int = 8192):
    """Exact argmin_k |x_n - e_k|^2 (lowest k on ties) for fp32 inputs, via fp64.

    Scores s = |e|^2 - 2 x.e in fp64 (row constant |x|^2 dropped); rows whose two best
    scores are closer than 1e-9 relative are re-ranked with explicit differences
    sum_i (x_i - e_i)^2 so duplicated codes tie exactly.
    """
    x64 = np.asarray(x, dtype=np.float64)
    e64 = np.asarray(codebook, dtype=np.float64)
    en = (e64 * e64).sum(axis=1)
    n = x64.shape[0]
    idx = np.empty(n, dtype=np.int64)
    gap = np.empty(n, dtype=np.float64)
    for s in range(0, n, chunk):
        xs = x64[s:s + chunk]
        sc = en[None, :] - 2.0 * (xs @ e64.T)
        i1 = sc.argmin(axis=1)
        best = sc[np.arange(len(xs)), i1]
        sc2 = sc.copy()
        sc2[np.arange(len(xs)), i1] = np.inf
        second = sc2.min(axis=1) if sc.shape[1] > 1 else np.full(len(xs), np.inf)
        g = second - best
        scale = np.abs(best) + (xs * xs).sum(axis=1) + 1e-300
        close = np.nonzero(g <= 1e-9 * scale)[0]
        for r in close:
            d = ((xs[r][None, :] - e64) ** 2).sum(axis=1)
            i1[r] = int(d.argmin())
            ds = np.sort(d)
            g[r] = ds[1] - ds[0] if len(ds) > 1 else np.inf
        idx[s:s + chunk] = i1
        gap[s:s + chunk] = g
    return (idx, gap) if return_gap else idx


def vq_forward(x_bchw: np.ndarray, weight: np.ndarray, codebook_mask=None, beta: float = 0.25):
    """Eval-mode VectorQuantize2.forward (quantize2_mask.py:157-191).

    x_bchw [B,D,H,W] fp32; weight [K+1,D] (row K is the padding row, excluded from search).
    codebook_mask [B,1,H,W] or None.  Returns (x_q [B,D,H,W], loss scalar fp32, idx [B,H,W]).
    Forward value of the straight-through output is x + (x_q - x).
    """
    x = np.asarray(x_bchw, dtype=np.float32)
    b, d, h, w = x.shape
    flat = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(-1, d)
    cb = np.asarray(weight, dtype=np.float32)[:-1]
    idx = argmin_exact(flat, cb)
    xq = cb[idx]
    diff2 = (xq - flat) ** 2
    if codebook_mask is not None:
        m = np.asarray(codebook_mask, dtype=np.float32).transpose(0, 2, 3, 1).reshape(-1, 1)
        diff2 = diff2 * m
    mse = diff2.mean(dtype=np.float64)
    loss = np.float32(beta * mse + mse)
    st = flat + (xq - flat)
    x_q = st.reshape(b, h, w, d).transpose(0, 3, 1, 2)
    return np.ascontiguousarray(x_q), loss, idx.reshape(b, h, w)


def ema_update(vectors: np.ndarray, idx: np.ndarray, cluster_size_ema: np.ndarray, embed_ema: np.ndarray,
               decay: float = 0.99, eps: float = 1e-5, restart_rows: np.ndarray | None = None):
    """One training-time codebook update (quantize2_mask.py:66-115), single process.

    vectors [N,D], idx [N]; returns (cluster_size_ema', embed_ema', weight[:K]').
    restart_rows: the K candidate replacement rows (reference: vectors[randperm][:K]); codes whose
    EMA count drops below 1 take the corresponding row and count 1.  None = restart disabled.
    """
    v = np.asarray(vectors, dtype=np.float32)
    k, d = embed_ema.shape
    counts = np.bincount(np.asarray(idx).reshape(-1), minlength=k).astype(np.float32)
    sums = np.zeros((k, d), dtype=np.float64)
    np.add.at(sums, np.asarray(idx).reshape(-1), v.astype(np.float64))
    sums = sums.astype(np.float32)
    n_ema = (cluster_size_ema.astype(np.float32) * np.float32(decay) + counts * np.float32(1 - decay)).astype(np.float32)
    s_ema = (embed_ema.astype(np.float32) * np.float32(decay) + sums * np.float32(1 - decay)).astype(np.float32)
    if restart_rows is not None:
        usage = (n_ema >= 1).astype(np.float32)
        s_ema = s_ema * usage[:, None] + np.asarray(restart_rows, dtype=np.float32)[:k] * (1 - usage)[:, None]
        n_ema = n_ema * usage + (1 - usage)
    n = n_ema.sum(dtype=np.float32)
    norm = n * (n_ema + np.float32(eps)) / (n + np.float32(k * eps))
    weight = (s_ema / norm[:, None]).astype(np.float32)
    return n_ema, s_ema, weight
