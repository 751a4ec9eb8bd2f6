#!/usr/bin/env python3
"""AttnBlock attention (one head of 256 channels over T tokens, B = 64): flash forward and the backward path, timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import _lib, kernels as K
dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
b, t, c = 64, int(os.environ.get("T", "1024")), 256
reps = int(os.environ.get("REPS", "10"))
q, k, v, do = (torch.randn(b * t, c, device=dev).to(torch.bfloat16) for _ in range(4))


def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


o, lse = K.attn_full_fwd(q, k, v, b, t, c ** -0.5)
fl = 4.0 * b * t * t * c
ms = timeit(lambda: K.attn_full_fwd(q, k, v, b, t, c ** -0.5))
print(f"attn_full_fwd  {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF/s")
ms = timeit(lambda: K.attn_full_bwd(q, k, v, o, do, lse, b, t, c ** -0.5))
print(f"attn_full_bwd (flash kernels) {ms:7.3f} ms {2.5 * fl / ms / 1e9:6.0f} TF/s")


def gemm_bwd():
    """the AttnBlock backward as layers.AttnBlock.bwd runs it by default (DVQ_ATTNBLOCK_BWD=gemm): probabilities recomputed, five batched GEMMs"""
    n, sc = t, c ** -0.5
    s = K.gemm_nt(q, k, n, n, c, c, c, n, batch=b, sa=n * c, sb=n * c, sc=n * n)
    p = K.softmax_rows(s, b * n, n, sc)
    dp = K.gemm_nt(do, v, n, n, c, c, c, n, batch=b, sa=n * c, sb=n * c, sc=n * n)
    dv32 = K.gemm_tn(p, do, n, n, c, n, c, c, batch=b, sa=n * n, sb=n * c, sc=n * c)
    ds = K.softmax_rows_bwd(p, dp, b * n, n, sc)
    kt = K.transpose(k, b, n, c)
    dq = K.gemm_nt(ds, kt, n, c, n, n, n, c, batch=b, sa=n * n, sb=c * n, sc=n * c)
    dk32 = K.gemm_tn(ds, q, n, n, c, n, c, c, batch=b, sa=n * n, sb=n * c, sc=n * c)
    return dq, dk32, dv32


try:
    ms = timeit(gemm_bwd)
    print(f"AttnBlock backward on batched GEMMs {ms:7.3f} ms {2.5 * fl / ms / 1e9:6.0f} TF/s")
except Exception as e:          # signature drift of the GEMM wrappers must not hide the numbers above
    print("gemm backward not timed:", type(e).__name__, e)
