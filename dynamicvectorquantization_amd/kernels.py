"""Thin tensor-level wrappers over the C ABI (no autograd here).

Every function takes/returns torch tensors living on a HIP device, passes raw device pointers and
the current stream to libdvq_hip.so and never synchronises.  Activations are NHWC contiguous
([N,H,W,C]); `dt(t)` maps torch dtypes to DVQ_F32 / DVQ_BF16.  torch is used for memory and streams
only -- a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib
from ._lib import ConvDesc, check

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def dt(t) -> int:
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d not in _DT:
        raise TypeError(f"unsupported dtype {d}; libdvq_hip computes in float32 or bfloat16")
    return _DT[d]


def vec(dtype) -> int:
    """channel granularity of the MFMA path (elements per 16 bytes)"""
    return 4 if dtype == torch.float32 else 8


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.DvqError("libdvq_hip kernels need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise _lib.DvqError("libdvq_hip kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def _praw(t):
    """device pointer of a tensor whose physical layout the caller vouches for (e.g. channel-last weight views)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.DvqError("libdvq_hip kernels need device tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def is_ohwi(t) -> bool:
    """True if a [Cout,Cin,KH,KW]-shaped tensor is physically stored [Cout][KH][KW][Cin] (FlatParams storage);
    False if it is torch-contiguous.  1x1 kernels are both: reported as contiguous."""
    if t.is_contiguous():
        return False
    co, ci, kh, kw = t.shape
    if t.stride() == (kh * kw * ci, 1, kw * ci, ci):
        return True
    raise _lib.DvqError(f"unsupported weight strides {t.stride()} for shape {tuple(t.shape)}")


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def lib():
    return _lib.load()


# ---------------------------------------------------------------------------------------------
# optional per-kernel-family timing with HIP events on the launch stream (bench.py roofline)
# ---------------------------------------------------------------------------------------------
_prof = None          # {family: [(start_event, end_event, flops, bytes), ...]} while profiling
_pool = []            # pre-created timing events (hipEventCreate is slow on some hosts: never create in the timed region)
_count = None         # launch counter (sizing pass)


def profiling_active() -> bool:
    """True while per-launch HIP-event timing or launch counting is on (such a step must run eagerly, not as a graph replay)"""
    return _prof is not None or _count is not None


def profile_count_start():
    """count timed launches without recording anything (used to size the event pool)"""
    global _count
    _count = 0


def profile_count_stop() -> int:
    global _count
    n, _count = _count or 0, None
    return n


def profile_prepare(max_launches):
    """create `max_launches` timing-event pairs (slow on some hosts: do it outside any timed region)"""
    global _pool
    _pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * max_launches)]
    for ev in _pool:          # torch creates the HIP event lazily at the first record(): force that now
        ev.record()
    torch.cuda.synchronize()


def profile_start(max_launches=0):
    """start recording with the prepared event pool (or create one now)"""
    global _prof
    if max_launches:
        profile_prepare(max_launches)
    _prof = {}


def profile_stop():
    """-> {family: dict(launches, ms, flops, bytes)}; synchronises."""
    global _prof, _pool
    p, _prof = _prof, None
    _pool = []
    torch.cuda.synchronize()
    out = {}
    for name, recs in (p or {}).items():
        out[name] = dict(launches=len(recs), ms=sum(s.elapsed_time(e) for s, e, _, _ in recs),
                         flops=sum(r[2] for r in recs), bytes=sum(r[3] for r in recs))
    return out


_shape_tags = os.environ.get("DVQ_PROFILE_SHAPES", "0") == "1"    # debug: split the families by call geometry
_cur_tag = [None]


def _tag(d, mode):
    if _shape_tags:
        _cur_tag[0] = f"{mode} N{d.N} {d.H}x{d.W} {d.Cin}->{d.Cout} k{d.KH}s{d.stride}{'u' if d.upsample else ''}"


def _timed(name, flops, nbytes, fn):
    global _count
    if _shape_tags and _cur_tag[0] is not None:
        name, _cur_tag[0] = f"{name} | {_cur_tag[0]}", None
    if _count is not None:
        _count += 1
    if _prof is None or len(_pool) < 2:
        return fn()
    s, e = _pool.pop(), _pool.pop()
    s.record()
    r = fn()
    e.record()
    _prof.setdefault(name, []).append((s, e, flops, nbytes))
    return r


def _halo_eligible(d: ConvDesc) -> bool:
    """mirrors halo_eligible()/dvq_conv3x3_halo_try in csrc: which kernel a conv call lands on (for timing labels)"""
    return (d.dtype == _lib.BF16 and d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad_t == 1 and d.pad_l == 1 and
            d.OH == d.H and d.OW == d.W and d.impl in (0, 4) and d.W % 32 == 0 and d.Cin % 64 == 0 and d.Cout % 8 == 0)


def _tn_family(d: ConvDesc, cin_real: int) -> str:
    """mirrors launch_tn() in csrc/igemm.hip: the kernel family a non-halo weight-gradient call lands on (timing labels)"""
    if d.dtype != _lib.BF16:
        return "igemm_tn_kernel"
    if (d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad_t == 0 and d.pad_l == 0 and not d.upsample and d.Cin >= 256 and d.Cout >= 256 and
            (d.impl != 0 or os.environ.get("DVQ_TN_1X1_PATCH", "1") == "0")):
        return "gemm_tn_wide_pipe_kernel"
    if d.Cin == 8 and 1 < d.KH * d.KW <= 16:
        return "igemm_tn_tr_kernel"          # thin: taps folded into the column tile
    if d.impl == 0 and os.environ.get("DVQ_CONV_TN_PATCH", "1") != "0":
        return "conv_tn_patch_kernel"
    return "igemm_tn_tr_kernel"


def _nt_family(d: ConvDesc, dgrad: bool) -> str:
    """mirrors launch_nt() in csrc/igemm.hip: the kernel family a non-halo forward / input-gradient call lands on (timing labels)"""
    cs, ncols = (d.Cout, d.Cin) if dgrad else (d.Cin, d.Cout)
    if d.dtype != _lib.BF16 or d.impl not in (0, 9):
        return "igemm_nt_glds_kernel"
    if (d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad_t == 0 and d.pad_l == 0 and not d.upsample and ncols >= 256 and cs % 64 == 0
            and d.impl == 0):
        return "gemm_nt_wide_pipe_kernel"
    if (cs % 64 == 0 and ncols % 8 == 0 and not d.upsample and d.KH * d.KW <= 16 and
            (not dgrad or d.stride == 1 or (d.H % 2 == 0 and d.W % 2 == 0))):
        return "conv_nt_pipe_kernel"
    return "igemm_nt_glds_kernel"


def _conv_cost(d: ConvDesc, esize: int):
    """algorithmic cost of one conv pass: 2*M*K*N flops; bytes = input + weights + output once each"""
    flops = 2 * d.N * d.OH * d.OW * d.Cout * d.KH * d.KW * d.Cin
    sh, sw = (d.H // 2, d.W // 2) if d.upsample else (d.H, d.W)
    nbytes = esize * (d.N * sh * sw * d.Cin + d.Cout * d.KH * d.KW * d.Cin + d.N * d.OH * d.OW * d.Cout)
    return flops, nbytes


# ---------------------------------------------------------------------------------------------
# VQ
# ---------------------------------------------------------------------------------------------
def vq_prepare(codebook: torch.Tensor) -> torch.Tensor:
    k, d = codebook.shape
    prep = torch.empty(lib().dvq_vq_prep_bytes(k, d), dtype=torch.uint8, device=codebook.device)
    check(lib().dvq_vq_prepare(_p(codebook), k, d, _p(prep), _s()), "dvq_vq_prepare")
    return prep


_vq_ws = {}


def _vq_workspace(n, device):
    """scratch of dvq_vq_argmin: its counters must be zero when a call starts and every call leaves them zero (the re-rank kernel
    re-arms them), so ONE zero-filled buffer per (device, stream, N) serves all calls on that stream -- no zero-fill launch per
    search.  Inside a stream capture a fresh zeroed buffer is used instead (it lives in the capture's pool; the fill is recorded
    with it)."""
    nbytes = lib().dvq_vq_argmin_workspace_bytes(n)
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, n)
    ws = _vq_ws.get(key)
    if ws is None:
        if len(_vq_ws) >= 16:
            _vq_ws.clear()
        ws = _vq_ws[key] = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    return ws


def vq_argmin(x: torch.Tensor, codebook: torch.Tensor, prep: torch.Tensor | None = None, impl: int = 0,
              return_flagged: bool = False):
    """x [N,D] (fp32/bf16), codebook [K,D] fp32 -> idx int64 [N] (exact argmin, lowest index on ties).
    return_flagged: also a device int32 [3] = rows of THIS call settled in fp64 {over all K codes (generic kernel only), over their
    candidate list / flagged residue classes, of those: rows with more candidate classes than an entry lists} -- a COPY made by a
    kernel on this stream (the counters live in scratch shared by every search of this (device, stream, N); under capture in a graph-pool
    buffer each replay refreshes)."""
    n, d = x.shape
    k = codebook.shape[0]
    assert codebook.dtype == torch.float32 and codebook.shape[1] == d
    if prep is None and impl != 1 and d in (64, 128, 256):
        prep = vq_prepare(codebook)
    idx = torch.empty(n, dtype=torch.int64, device=x.device)
    ws = _vq_workspace(n, x.device)
    nbytes = n * d * x.element_size() + k * d * 4 + n * 8
    try:
        _timed("vq_argmin", 2 * n * k * d, nbytes, lambda: check(
            lib().dvq_vq_argmin(_p(x), dt(x), _p(codebook), _p(prep), n, k, d, _p(idx), _p(ws), impl, _s()), "dvq_vq_argmin"))
    except Exception:
        _vq_ws.clear()            # a failed call may leave the counters armed
        raise
    if return_flagged:
        return idx, ws[16:28].view(torch.int32).mul(1)
    return idx


def vq_distances(x, codebook, out=None):
    """x [N,D] (fp32/bf16), codebook [K,D] fp32 -> fp32 [N,K] squared distances (the reference's addmm formula); analysis API"""
    n, d = x.shape
    k = codebook.shape[0]
    if out is None:
        out = torch.empty(n, k, dtype=torch.float32, device=x.device)
    step = 1 << 19
    for a in range(0, n, step):
        b = min(n, a + step)
        check(lib().dvq_vq_distances(_p(x[a:b]), dt(x), _p(codebook), b - a, k, d, _p(out[a:b]), _s()), "dvq_vq_distances")
    return out


def vq_gather_loss(x, codebook, idx, mask=None):
    n, d = x.shape
    xq = torch.empty_like(x)
    loss_sum = torch.zeros(1, dtype=torch.float64, device=x.device)
    check(lib().dvq_vq_gather_loss(_p(x), dt(x), _p(codebook), _p(idx), _p(mask), n, d, _p(xq), _p(loss_sum), _s()),
          "dvq_vq_gather_loss")
    return xq, loss_sum


def vq_backward(g_xq, x, codebook, idx, mask, coef_dev):
    n, d = x.shape
    dx = torch.empty_like(x)
    check(lib().dvq_vq_backward(_p(g_xq), _p(x), dt(x), _p(codebook), _p(idx), _p(mask), _p(coef_dev), n, d, _p(dx), _s()),
          "dvq_vq_backward")
    return dx


def vq_embed(codebook, idx, dtype=torch.float32, out=None):
    d = codebook.shape[1]
    flat = idx.reshape(-1).contiguous()
    if out is None:
        out = torch.empty(flat.numel(), d, dtype=dtype, device=codebook.device)
    check(lib().dvq_vq_embed(_p(codebook), _p(flat), flat.numel(), d, dt(dtype), _p(out), _s()), "dvq_vq_embed")
    return out.reshape(*idx.shape, d)


def vq_ema_stats(x, idx, k, out=None):
    n, d = x.shape
    stats = torch.empty(k, d + 1, dtype=torch.float32, device=x.device) if out is None else out
    check(lib().dvq_vq_ema_stats(_p(x), dt(x), _p(idx), n, k, d, _p(stats), _s()), "dvq_vq_ema_stats")
    return stats


def vq_ema_apply(stats, restart_rows, decay, eps, cluster_size_ema, embed_ema, weight):
    k, d = embed_ema.shape
    check(lib().dvq_vq_ema_apply(_p(stats), _p(restart_rows), decay, eps, k, d, _p(cluster_size_ema), _p(embed_ema),
                                 _p(weight), None, _s()), "dvq_vq_ema_apply")


# ---------------------------------------------------------------------------------------------
# entropy / gate
# ---------------------------------------------------------------------------------------------
def patch_entropy_gate(img: torch.Tensor, patch: int, threshold: float | None, bins=(-1.0, 1.0)):
    """img NCHW fp32 [B,3,H,W] -> (entropy [B,h,w] fp32, gate int64 [B,h,w,2] or None); bins = range of the 32 histogram bins"""
    b, c, h, w = img.shape
    assert c == 3 and img.dtype == torch.float32
    ent = torch.empty(b, h // patch, w // patch, dtype=torch.float32, device=img.device)
    gate = None
    if threshold is not None:
        gate = torch.empty(b, h // patch, w // patch, 2, dtype=torch.int64, device=img.device)
    if tuple(bins) == (-1.0, 1.0):
        check(lib().dvq_patch_entropy_gate(_p(img), b, h, w, patch, float(threshold or 0.0), _p(ent), _p(gate), _s()),
              "dvq_patch_entropy_gate")
    else:
        check(lib().dvq_patch_entropy_gate_range(_p(img), b, h, w, patch, float(bins[0]), float(bins[1]), float(threshold or 0.0),
                                                 _p(ent), _p(gate), _s()), "dvq_patch_entropy_gate_range")
    return ent, gate


# ---------------------------------------------------------------------------------------------
# GroupNorm (+swish)
# ---------------------------------------------------------------------------------------------
def gn_stats(x, groups=32):
    """x NHWC -> fp64 [N,G,2] (sum, sum of squares)"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    stats = zeros_small((n, groups, 2), torch.float64, x.device)
    check(lib().dvq_gn_stats(_p(x), dt(x), n, hw, c, groups, _p(stats), _s()), "dvq_gn_stats")
    return stats


def gn_forward(x, gamma, beta, groups=32, eps=1e-6, silu=True, stats=None):
    """x NHWC [N,H,W,C]; returns (y, mean_rstd [N,G,2] fp32).  `stats` may come from a producer's fused epilogue."""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    if stats is None:
        stats = gn_stats(x, groups)
    y = torch.empty_like(x)
    mr = torch.empty(n, groups, 2, dtype=torch.float32, device=x.device)
    check(lib().dvq_gn_apply(_p(x), dt(x), n, hw, c, groups, eps, _p(stats), _p(gamma), _p(beta), int(silu), _p(y), _p(mr),
                             _s()), "dvq_gn_apply")
    return y, mr


def gn_scale_shift(x, gamma, beta, groups=32, eps=1e-6, stats=None):
    """per-(n,c) {scale, shift} fp32 [N,C,2] for the fused conv prologue + mean_rstd [N,G,2] for the backward"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    if stats is None:
        stats = gn_stats(x, groups)
    ss = torch.empty(n, c, 2, dtype=torch.float32, device=x.device)
    mr = torch.empty(n, groups, 2, dtype=torch.float32, device=x.device)
    check(lib().dvq_gn_scale_shift(_p(stats), _p(gamma), _p(beta), n, hw, c, groups, eps, _p(ss), _p(mr), _s()),
          "dvq_gn_scale_shift")
    return ss, mr


_GN_PARTIALS = os.environ.get("DVQ_GN_PARTIALS", "1") != "0"


def gn_backward(x, dy, mean_rstd, gamma, beta, dgamma, dbeta, groups=32, silu=True, addend=None, fixed_stats=False):
    """dgamma/dbeta (fp32 [C]) are accumulated into; returns dx.  fixed_stats: mean / rstd are constants, not functions of x (ActNorm:
    a per-channel affine): the statistic terms of dx are dropped by zeroing the reduced sums, dx = rstd * gamma * dz"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    red = zeros_small((n, groups, 2), torch.float64, x.device)
    part = None
    if _GN_PARTIALS:      # per-block partials + fold kernel instead of thousands of atomics per address
        part = torch.empty(lib().dvq_gn_bwd_partial_bytes(n, hw, c), dtype=torch.uint8, device=x.device)
    check(lib().dvq_gn_bwd_reduce(_p(x), _p(dy), dt(x), n, hw, c, groups, _p(mean_rstd), _p(gamma), _p(beta), int(silu),
                                  _p(red), _p(dgamma), _p(dbeta), _p(part), _s()), "dvq_gn_bwd_reduce")
    if fixed_stats:
        red.zero_()
    dx = torch.empty_like(x)
    check(lib().dvq_gn_bwd_dx(_p(x), _p(dy), dt(x), n, hw, c, groups, _p(mean_rstd), _p(gamma), _p(beta), int(silu),
                              _p(red), _p(addend), _p(dx), _s()), "dvq_gn_bwd_dx")
    return dx


# ---------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------
def conv_desc(n, h, w, cin, cout, kh, kw, stride, pad_t, pad_l, oh, ow, upsample, dtype, impl=0) -> ConvDesc:
    return ConvDesc(n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l, int(upsample), dt(dtype), impl)


def pack_weight(master_oihw, cin_p, cout_p, dtype, want_w=True, want_wt=True):
    cout, cin, kh, kw = master_oihw.shape
    dev = master_oihw.device
    w = torch.empty(cout, kh, kw, cin_p, dtype=dtype, device=dev) if want_w else None
    wt = torch.zeros(cin_p, kh, kw, cout_p, dtype=dtype, device=dev) if want_wt else None   # rows >= Cin stay zero
    check(lib().dvq_pack_weight(_p(master_oihw), cout, cin, kh, kw, cin_p, cout_p, dt(dtype), _p(w), _p(wt), _s()),
          "dvq_pack_weight")
    return w, wt


def pack_weight_into(master, cin_p, cout_p, dtype, w, wt):
    cout, cin, kh, kw = master.shape
    flag = dt(dtype) | (256 if is_ohwi(master) else 0)
    check(lib().dvq_pack_weight(_praw(master), cout, cin, kh, kw, cin_p, cout_p, flag, _p(w), _p(wt), _s()),
          "dvq_pack_weight")


def unpack_wgrad(dw, grad_oihw, cin_p):
    cout, cin, kh, kw = grad_oihw.shape
    check(lib().dvq_unpack_wgrad(_p(dw), cout, cin, kh, kw, cin_p, _p(grad_oihw), _s()), "dvq_unpack_wgrad")


def conv_fused_ok(d: ConvDesc) -> bool:
    return bool(lib().dvq_conv3x3_fused_ok(C.byref(d)))


ACT_NONE, ACT_SWISH, ACT_LRELU, ACT_RELU = 0, 1, 2, 3      # include/dvq_hip.h DVQ_ACT_*


def _x3_halo(d: ConvDesc, t, dgrad: bool) -> bool:
    """fp32x3 forward / input gradient of this call on the halo kernel (bf16 planes on the channel axis, fp32 output)?
    DVQ_X3_HALO=0: stay on the fp32 kernel that splits at every fragment read (A/B switch)"""
    return (t.dtype == torch.float32 and d.KH == 3 and d.stride == 1 and os.environ.get("DVQ_X3_HALO", "1") != "0" and
            bool(lib().dvq_fp32_split()) and bool(lib().dvq_conv3x3_x3_ok(C.byref(d), int(dgrad))))


def conv2d_fwd(d: ConvDesc, x, w, bias, residual=None, gn_ss=None, out_stats=None, out_groups=0, act=ACT_NONE):
    y = torch.empty(d.N, d.OH, d.OW, d.Cout, dtype=x.dtype, device=x.device)
    fl, nb = _conv_cost(d, x.element_size())
    if gn_ss is None and out_stats is None and _x3_halo(d, x, False):
        _tag(d, "fwd x3 halo" + ("+res" if residual is not None else "") + ("+act" if act != ACT_NONE else ""))
        need = int(lib().dvq_conv3x3_x3_scratch_bytes(C.byref(d), 0))
        scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
        _timed("conv_x3_halo", fl, nb, lambda: check(
            lib().dvq_conv2d_fwd_x3(C.byref(d), _p(x), _p(w), _p(bias), _p(residual), _p(y), act, _p(scratch), need, _s()),
            "dvq_conv2d_fwd_x3"))
        return y
    if _shape_tags:       # per-shape tables split the forward by epilogue / prologue variant (tools/debug/step_shapes.py)
        _tag(d, "fwd" + ("+res" if residual is not None else "") + ("+gn" if gn_ss is not None else "") +
             ("+st" if out_stats is not None else "") + ("+act" if act != ACT_NONE else ""))
    if out_stats is not None:
        ensure_workspace(x.device)        # per-tile statistics partials of the halo kernel
    if act != ACT_NONE:
        assert residual is None and gn_ss is None and out_stats is None
        _timed("conv3x3_halo_kernel" if _halo_eligible(d) and d.H % 8 == 0 else _nt_family(d, False), fl, nb, lambda: check(
            lib().dvq_conv2d_fwd_act(C.byref(d), _p(x), _p(w), _p(bias), _p(y), act, _s()), "dvq_conv2d_fwd_act"))
        return y
    if gn_ss is not None or out_stats is not None:
        _timed("conv3x3_halo_kernel", fl, nb, lambda: check(
            lib().dvq_conv2d_fwd_ex(C.byref(d), _p(x), _p(w), _p(bias), _p(residual), _p(y), _p(gn_ss), _p(out_stats),
                                    out_groups, _s()), "dvq_conv2d_fwd_ex"))
        return y
    _timed("conv3x3_halo_kernel" if _halo_eligible(d) and d.H % 8 == 0 else _nt_family(d, False), fl, nb, lambda: check(
        lib().dvq_conv2d_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(residual), _p(y), _s()), "dvq_conv2d_fwd"))
    return y


def conv2d_dgrad(d: ConvDesc, dy, wt, mask=None, mask_act=ACT_NONE):
    """mask: output of the activation that produced this conv's input; the gradient is gated through it"""
    sh, sw = (d.H // 2, d.W // 2) if d.upsample else (d.H, d.W)
    dx = torch.empty(d.N, sh, sw, d.Cin, dtype=dy.dtype, device=dy.device)
    ws = torch.empty(d.N, d.H, d.W, d.Cin, dtype=dy.dtype, device=dy.device) if d.upsample else None
    fl, nb = _conv_cost(d, dy.element_size())
    if _x3_halo(d, dy, True):
        _tag(d, "dgrad x3 halo")
        need = int(lib().dvq_conv3x3_x3_scratch_bytes(C.byref(d), 1))
        scratch = torch.empty(need, dtype=torch.uint8, device=dy.device)
        _timed("conv_x3_halo", fl, nb, lambda: check(
            lib().dvq_conv2d_dgrad_x3(C.byref(d), _p(dy), _p(wt), _p(dx), _p(ws), _p(mask), mask_act, _p(scratch), need, _s()),
            "dvq_conv2d_dgrad_x3"))
        return dx
    _tag(d, "dgrad")
    _timed("conv3x3_halo_kernel" if _halo_eligible(d) and d.H % 8 == 0 and d.Cout % 64 == 0 else _nt_family(d, True), fl, nb,
           lambda: check(
        lib().dvq_conv2d_dgrad_mask(C.byref(d), _p(dy), _p(wt), _p(dx), _p(ws), _p(mask), mask_act, _s()),
        "dvq_conv2d_dgrad_mask"))
    return dx


def conv2d_wgrad(d: ConvDesc, x, dy, db=None):
    """returns dw (fp32 packed [Cout,KH,KW,Cin]); db (fp32 [Cout]) is accumulated into when given"""
    dw = torch.zeros(d.Cout, d.KH, d.KW, d.Cin, dtype=torch.float32, device=x.device)
    fl, nb = _conv_cost(d, x.element_size())
    _timed("conv_wgrad", fl, nb, lambda: check(
        lib().dvq_conv2d_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _s()), "dvq_conv2d_wgrad"))
    return dw


_workspace = {}
# 8 per-stream slots (csrc/misc.hip: dvq_workspace_stream) of 80 MB: 256 workgroups x (9 x 128 x 64 + 128) fp32 partials of the
# split-K weight-gradient kernels
WORKSPACE_BYTES = 8 * (80 << 20)


def workspace_release(stream):
    """give back the scratch slot a (capture) stream holds -- called when its recording is dropped"""
    if _workspace:
        lib().dvq_workspace_release(C.c_void_p(int(stream.cuda_stream)))


def copy_kernel_(dst, src):
    """dst <- src through an elementwise KERNEL.  torch's copy_ between contiguous same-dtype tensors is a hipMemcpyAsync, which a
    recorded step keeps as a 1-D memcpy node -- the one node kind a launch list (csrc/cmdlist.hip) cannot re-issue.  x * 1 is exact."""
    return torch.mul(src, 1, out=dst)


class CmdList:
    """launch list of one captured segment (csrc/cmdlist.hip); `graph` is the torch.cuda.CUDAGraph(keep_graph=True) whose nodes
    own the argument blocks -- kept alive here, released after the list"""

    def __init__(self, graph):
        self.graph = graph
        h = C.c_void_p()
        check(lib().dvq_cmdlist_create(C.c_void_p(int(graph.raw_cuda_graph())), C.byref(h)), "dvq_cmdlist_create")
        self.handle = h
        info = (C.c_int64 * 4)()
        check(lib().dvq_cmdlist_info(h, info), "dvq_cmdlist_info")
        self.kernels, self.side_kernels, self.waits = int(info[0]), int(info[1]), int(info[2])
        self.other, self.side_open = int(info[3]) & 0xffffffff, bool(int(info[3]) >> 32)

    def replay(self, main, side):
        check(lib().dvq_cmdlist_replay(self.handle, C.c_void_p(int(main.cuda_stream)), C.c_void_p(int(side.cuda_stream))),
              "dvq_cmdlist_replay")

    def __del__(self):
        try:
            if self.handle:
                lib().dvq_cmdlist_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
        self.graph = None


def ensure_workspace(device):
    """register the process-wide scratch buffer of libdvq_hip (kept alive here)"""
    device = torch.device(device)
    if _workspace.get("dev") != device and os.environ.get("DVQ_NO_WORKSPACE", "0") != "1":
        buf = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        check(lib().dvq_set_workspace(buf.data_ptr(), buf.numel()), "dvq_set_workspace")
        _workspace.update(dev=device, buf=buf)


def _wgrad_x3_planes() -> bool:
    """DVQ_X3_WGRAD_PLANES=0: fp32x3 weight gradients stay on the fp32 kernel that splits at every fragment read (A/B switch)"""
    return os.environ.get("DVQ_X3_WGRAD_PLANES", "1") != "0"


def split_bf16_planes(x2d, cout=None):
    """fp32 [rows, C] -> (hi, lo) bf16 [rows, cout]: hi = RNE(x), lo = RNE(x - hi); channels >= C are zero"""
    rows, c = x2d.shape
    cout = c if cout is None else cout
    hi = torch.empty(rows, cout, dtype=torch.bfloat16, device=x2d.device)
    lo = torch.empty_like(hi)
    check(lib().dvq_split_bf16_planes(_p(x2d), _p(hi), _p(lo), rows, c, cout, _s()), "dvq_split_bf16_planes")
    return hi, lo


def conv2d_wgrad_oihw(d: ConvDesc, x, dy, cin_real, cout_real, grad_oihw, db=None, gn_ss=None):
    """accumulate the weight gradient straight into the [Cout,Cin,KH,KW] fp32 grad (and db into [Cout]);
    gn_ss: the fused GroupNorm+swish of the forward is re-applied to x inside the kernel"""
    fl, nb = _conv_cost(d, x.element_size())
    ensure_workspace(x.device)
    if gn_ss is None and _wgrad_x3_planes() and x.dtype == torch.float32 and d.impl == 0 and bool(lib().dvq_fp32_split()):
        # fp32x3: bf16 planes of both operands (scratch from the caching allocator), three launches of the bf16 weight-gradient kernels
        _tag(d, "wgrad x3 planes")
        need = int(lib().dvq_conv2d_wgrad_x3_scratch_bytes(C.byref(d)))
        scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
        _timed("conv_wgrad_x3_planes", fl, nb, lambda: check(
            lib().dvq_conv2d_wgrad_oihw_x3(C.byref(d), _p(x), _p(dy), cin_real, cout_real, _praw(grad_oihw), _p(db),
                                           int(is_ohwi(grad_oihw)), _p(scratch), need, _s()), "dvq_conv2d_wgrad_oihw_x3"))
        return
    _tag(d, "wgrad")
    if gn_ss is not None:
        _timed("conv3x3_halo_wgrad_kernel", fl, nb, lambda: check(
            lib().dvq_conv2d_wgrad_oihw_ex(C.byref(d), _p(x), _p(dy), cin_real, cout_real, _praw(grad_oihw), _p(db),
                                           int(is_ohwi(grad_oihw)), _p(gn_ss), _s()), "dvq_conv2d_wgrad_oihw_ex"))
        return
    _timed("conv3x3_halo_wgrad_kernel" if _halo_eligible(d) and d.H % 4 == 0 else _tn_family(d, cin_real), fl, nb, lambda: check(
        lib().dvq_conv2d_wgrad_oihw(C.byref(d), _p(x), _p(dy), cin_real, cout_real, _praw(grad_oihw), _p(db),
                                    int(is_ohwi(grad_oihw)), _s()), "dvq_conv2d_wgrad_oihw"))


def set_deterministic(on: bool):
    """opt-in: weight-gradient split reductions run unsplit or through partials + a fold kernel (bit-reproducible gradients)"""
    check(lib().dvq_set_deterministic(int(bool(on))), "dvq_set_deterministic")


def deterministic() -> bool:
    return bool(lib().dvq_deterministic())


def pack_weights_multi(table_dev, n_entries, total_work):
    check(lib().dvq_pack_weights_multi(_p(table_dev), n_entries, total_work, _s()), "dvq_pack_weights_multi")


def linear_pack_multi(table_dev, n_entries, total_tiles):
    check(lib().dvq_linear_pack_multi(_p(table_dev), n_entries, total_tiles, _s()), "dvq_linear_pack_multi")


# ---------------------------------------------------------------------------------------------
# zero arena: one memset per step instead of one torch.zeros per small statistics buffer
# ---------------------------------------------------------------------------------------------
_arena = {"buf": None, "off": 0}


def arena_reset(device=None, nbytes=8 << 20):
    a = _arena
    if a["buf"] is None or (device is not None and a["buf"].device != torch.device(device)):
        a["buf"] = torch.zeros(nbytes, dtype=torch.uint8, device=device or "cuda")
    else:
        a["buf"].zero_()
    a["off"] = 0


def zeros_small(shape, dtype, device):
    """zero-initialised small buffer: a slice of the per-step arena when one is active, else torch.zeros"""
    a = _arena
    n = 1
    for s_ in shape:
        n *= int(s_)
    nb = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
    if a["buf"] is None or a["buf"].device != torch.device(device) or a["off"] + nb > a["buf"].numel():
        return torch.zeros(shape, dtype=dtype, device=device)
    out = a["buf"][a["off"]:a["off"] + nb].view(dtype)[:n].view(shape)
    a["off"] += nb
    return out


def nchw_to_nhwc_pad(img, cp, dtype):
    b, c, h, w = img.shape
    out = torch.empty(b, h, w, cp, dtype=dtype, device=img.device)
    check(lib().dvq_nchw_to_nhwc_pad(_p(img), b, c, h, w, cp, dt(dtype), _p(out), _s()), "dvq_nchw_to_nhwc_pad")
    return out


def nhwc_pad_to_nchw(x, c):
    b, h, w, cp = x.shape
    out = torch.empty(b, c, h, w, dtype=torch.float32, device=x.device)
    check(lib().dvq_nhwc_pad_to_nchw(_p(x), dt(x), b, c, h, w, cp, _p(out), _s()), "dvq_nhwc_pad_to_nchw")
    return out


# ---------------------------------------------------------------------------------------------
# GEMMs / attention pieces
# ---------------------------------------------------------------------------------------------
def probe_mfma_rate(random_operands=True):
    """(TFLOP/s, shader MHz) a register-only bf16 MFMA loop sustains on the current device (diagnostics for bench.py)"""
    tf, mhz = C.c_float(0.0), C.c_float(0.0)
    check(lib().dvq_probe_mfma_rate(int(bool(random_operands)), C.byref(tf), C.byref(mhz), _s()), "dvq_probe_mfma_rate")
    return float(tf.value), float(mhz.value)


def gemm_nt(a, b, m, n, k, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0, alpha=1.0, bias=None, bias_mode=0, out=None,
            impl=0, residual=None):
    """C[b][m][n] = alpha * sum_k A[b][m][k] B[b][n][k] (+bias) (+residual, same layout as C).  a, b, out are flat device tensors."""
    if out is None:
        out = torch.empty(batch * m * ldc if sc == 0 else batch * sc, dtype=a.dtype, device=a.device)
    es = a.element_size()
    if residual is not None:
        _timed("gemm_nt", 2 * batch * m * n * k, es * batch * (m * k + n * k + 2 * m * n), lambda: check(
            lib().dvq_gemm_nt_res(_p(a), _p(b), _p(out), _p(residual), dt(a), m, n, k, lda, ldb, ldc, batch, sa, sb, sc, alpha, _p(bias),
                                  bias_mode, _s()), "dvq_gemm_nt_res"))
        return out
    _timed("gemm_nt", 2 * batch * m * n * k, es * batch * (m * k + n * k + m * n), lambda: check(
        lib().dvq_gemm_nt(_p(a), _p(b), _p(out), dt(a), m, n, k, lda, ldb, ldc, batch, sa, sb, sc, alpha, _p(bias),
                          bias_mode, impl, _s()), "dvq_gemm_nt"))
    return out


def gemm_tn(a, b, mred, i, j, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0, out=None, impl=0, colsum=None):
    """C[b][i][j] (fp32) += sum_m A[b][m][i] B[b][m][j]; colsum (fp32 [i], batch 1): += column sums of A from the same pass"""
    if out is None:
        out = torch.zeros(batch * (sc if sc else i * ldc), dtype=torch.float32, device=a.device)
    es = a.element_size()
    if a.is_cuda and es == 2 and mred >= 1024 and i >= 256 and j >= 256:
        ensure_workspace(a.device)        # split partials of the 256 x 256 kernel (fold kernel instead of fp32 atomics)
    if colsum is not None:
        assert batch == 1
        _timed("gemm_tn", 2 * mred * i * j, es * mred * (i + j) + 4 * i * j, lambda: check(
            lib().dvq_gemm_tn_colsum(_p(a), _p(b), _p(out), _p(colsum), dt(a), mred, i, j, lda, ldb, ldc, impl, _s()), "dvq_gemm_tn_colsum"))
        return out
    _timed("gemm_tn", 2 * batch * mred * i * j, es * batch * mred * (i + j) + 4 * batch * i * j, lambda: check(
        lib().dvq_gemm_tn(_p(a), _p(b), _p(out), dt(a), mred, i, j, lda, ldb, ldc, batch, sa, sb, sc, impl, _s()),
        "dvq_gemm_tn"))
    return out


def softmax_rows(s, rows, length, scale):
    p = torch.empty_like(s)
    check(lib().dvq_softmax_rows(_p(s), dt(s), rows, length, scale, _p(p), _s()), "dvq_softmax_rows")
    return p


def softmax_rows_bwd(p, dp, rows, length, scale):
    ds = torch.empty_like(p)
    check(lib().dvq_softmax_rows_bwd(_p(p), _p(dp), dt(p), rows, length, scale, _p(ds), _s()), "dvq_softmax_rows_bwd")
    return ds


def transpose(x, batch, r, c):
    out = torch.empty_like(x)
    check(lib().dvq_transpose(_p(x), dt(x), batch, r, c, _p(out), _s()), "dvq_transpose")
    return out


# ---------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------
def dual_merge(h_fine, h_coarse, grain):
    b, h, w, c = h_coarse.shape
    out = torch.empty_like(h_fine)
    mask = torch.empty(b, 2 * h, 2 * w, dtype=torch.float32, device=h_fine.device)
    check(lib().dvq_dual_merge(_p(h_fine), _p(h_coarse), _p(grain), dt(h_fine), b, h, w, c, _p(out), _p(mask), _s()),
          "dvq_dual_merge")
    return out, mask


def dual_merge_bwd(g_dual, grain):
    b, h2, w2, c = g_dual.shape
    h, w = h2 // 2, w2 // 2
    gf = torch.empty_like(g_dual)
    gc = torch.empty(b, h, w, c, dtype=g_dual.dtype, device=g_dual.device)
    check(lib().dvq_dual_merge_bwd(_p(g_dual), _p(grain), dt(g_dual), b, h, w, c, _p(gf), _p(gc), _s()),
          "dvq_dual_merge_bwd")
    return gf, gc


def add(a, b):
    y = torch.empty_like(a)
    check(lib().dvq_add(_p(a), _p(b), dt(a), a.numel(), _p(y), _s()), "dvq_add")
    return y


def channel_shift_add8(a, b, shift):
    """a, b: [..., 8] bf16 pixels -> a + (b moved up by `shift` channels)"""
    assert a.shape == b.shape and a.shape[-1] == 8 and a.dtype == torch.bfloat16
    y = torch.empty_like(a)
    check(lib().dvq_channel_shift_add8(_p(a), _p(b), dt(a), a.numel() // 8, int(shift), _p(y), _s()), "dvq_channel_shift_add8")
    return y


def add_bias_bcast(x, bias):
    batch = x.shape[0]
    y = torch.empty_like(x)
    check(lib().dvq_add_bias_bcast(_p(x), _p(bias), dt(x), batch, x.numel() // batch, _p(y), _s()), "dvq_add_bias_bcast")
    return y


def sum_batch(x, out):
    batch = x.shape[0]
    check(lib().dvq_sum_batch(_p(x), dt(x), batch, x.numel() // batch, _p(out), _s()), "dvq_sum_batch")
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().dvq_cast(_p(x), dt(x), _p(out), dt(dtype), x.numel(), _s()), "dvq_cast")
    return out


def l1_loss(x, xrec, scale_dev=None, want_grad=False):
    loss_sum = torch.zeros(1, dtype=torch.float64, device=x.device)
    g = torch.empty_like(xrec) if want_grad else None
    check(lib().dvq_l1_loss(_p(x), _p(xrec), x.numel(), _p(loss_sum), _p(scale_dev), _p(g), _s()), "dvq_l1_loss")
    return loss_sum, g


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    check(lib().dvq_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, _s()), "dvq_adam")


def fill(p, value):
    check(lib().dvq_fill_f32(_p(p), float(value), p.numel(), _s()), "dvq_fill_f32")


# ---------------------------------------------------------------------------------------------
# loss networks (LPIPS / PatchGAN)
# ---------------------------------------------------------------------------------------------
def affine_channels(x, a, b=None):
    """y[..., c] = x[..., c] * a[c] + b[c]"""
    y = torch.empty_like(x)
    check(lib().dvq_affine_channels(_p(x), dt(x), x.numel(), x.shape[-1], _p(a), _p(b), _p(y), _s()), "dvq_affine_channels")
    return y


def axpy_dev(a, b, scale_dev):
    """a + scale_dev[0] * b (scale_dev: fp32 device scalar)"""
    y = torch.empty_like(a)
    check(lib().dvq_axpy_dev(_p(a), _p(b), _p(scale_dev), dt(a), a.numel(), _p(y), _s()), "dvq_axpy_dev")
    return y


def maxpool2x2(x):
    n, h2, w2, c = x.shape
    y = torch.empty(n, h2 // 2, w2 // 2, c, dtype=x.dtype, device=x.device)
    check(lib().dvq_maxpool2x2(_p(x), dt(x), n, h2 // 2, w2 // 2, c, _p(y), _s()), "dvq_maxpool2x2")
    return y


def maxpool2x2_relu_bwd(a, dpool=None, dtap=None):
    n, h2, w2, c = a.shape
    dz = torch.empty_like(a)
    check(lib().dvq_maxpool2x2_relu_bwd(_p(a), _p(dpool), _p(dtap), dt(a), n, h2 // 2, w2 // 2, c, _p(dz), _s()),
          "dvq_maxpool2x2_relu_bwd")
    return dz


def lpips_head(f0, f1, lin, val, gscale=0.0, want_grad=False, p_drop=0.0, seed=0):
    """val[n] (fp32, accumulated) += LPIPS term of one tap; returns d val / d f1 * gscale (or None).  p_drop > 0: NetLinLayer's
    dropout on the squared differences (hash-seeded mask, the same in the value and in the gradient)"""
    n, c = f0.shape[0], f0.shape[-1]
    hw = f0.numel() // (n * c)
    df1 = torch.empty_like(f1) if want_grad else None
    check(lib().dvq_lpips_head_drop(_p(f0), _p(f1), _p(lin), dt(f0), n, hw, c, _p(val), float(gscale), _p(df1), float(p_drop), int(seed),
                                    _s()), "dvq_lpips_head_drop")
    return df1


# ---------------------------------------------------------------------------------------------
# feature-routed (Gumbel) dual / triple grain pieces
# ---------------------------------------------------------------------------------------------
def avgpool_slice(x, k, out, coff):
    """mean over k x k windows of x [N,h*k,w*k,C] -> channel slice [coff, coff+C) of out [N,h,w,ldy]"""
    n, hk, wk, c = x.shape
    check(lib().dvq_avgpool_slice(_p(x), dt(x), n, hk // k, wk // k, c, k, _p(out), out.shape[-1], coff, _s()), "dvq_avgpool_slice")


def avgpool_slice_bwd(dy, coff, c, k):
    n, h, w, ldy = dy.shape
    dx = torch.empty(n, h * k, w * k, c, dtype=dy.dtype, device=dy.device)
    check(lib().dvq_avgpool_slice_bwd(_p(dy), dt(dy), ldy, coff, n, h, w, c, k, _p(dx), _s()), "dvq_avgpool_slice_bwd")
    return dx


def silu(x):
    y = torch.empty_like(x)
    check(lib().dvq_silu(_p(x), dt(x), x.numel(), _p(y), _s()), "dvq_silu")
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().dvq_silu_bwd(_p(x), _p(dy), dt(x), x.numel(), _p(dx), _s()), "dvq_silu_bwd")
    return dx


def relu(x):
    y = torch.empty_like(x)
    check(lib().dvq_relu(_p(x), dt(x), x.numel(), _p(y), _s()), "dvq_relu")
    return y


def relu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().dvq_relu_bwd(_p(x), _p(dy), dt(x), x.numel(), _p(dx), _s()), "dvq_relu_bwd")
    return dx


def upsample_nearest2x(x):
    """NHWC [N,h,w,C] -> [N,2h,2w,C] (F.interpolate(scale_factor=2, mode="nearest"))"""
    n, h, w, c = x.shape
    y = torch.empty(n, 2 * h, 2 * w, c, dtype=x.dtype, device=x.device)
    check(lib().dvq_upsample_nearest2x(_p(x), dt(x), n, h, w, c, _p(y), _s()), "dvq_upsample_nearest2x")
    return y


def upsample_nearest2x_bwd(dy):
    n, h2, w2, c = dy.shape
    dx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=dy.dtype, device=dy.device)
    check(lib().dvq_upsample_nearest2x_bwd(_p(dy), dt(dy), n, h2 // 2, w2 // 2, c, _p(dx), _s()), "dvq_upsample_nearest2x_bwd")
    return dx


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    for t in tensors:
        _p(t)                 # device / contiguity checks
    return arr


def grain_merge(heads, idx, scale=None):
    """heads: [coarsest, ..., finest] NHWC; idx int64 [N,hc,wc]; scale fp32 [N,hc,wc] or None -> (merged, codebook mask)"""
    n, hc, wc, c = heads[0].shape
    f = 1 << (len(heads) - 1)
    out = torch.empty(n, hc * f, wc * f, c, dtype=heads[0].dtype, device=heads[0].device)
    mask = torch.empty(n, hc * f, wc * f, dtype=torch.float32, device=out.device)
    arr = _ptr_array(heads)
    check(lib().dvq_grain_merge(arr, len(heads), _p(idx), _p(scale), dt(out), n, hc, wc, c, _p(out), _p(mask), _s()),
          "dvq_grain_merge")
    return out, mask


def grain_merge_bwd(g_out, heads, idx, scale=None, want_dscale=False):
    """-> ([d head_l], d scale or None)"""
    n, hc, wc, c = heads[0].shape
    dheads = [torch.empty_like(h) for h in heads]
    dscale = torch.empty(n, hc, wc, dtype=torch.float32, device=g_out.device) if want_dscale else None
    check(lib().dvq_grain_merge_bwd(_p(g_out), _ptr_array(heads), len(heads), _p(idx), _p(scale), dt(g_out), n, hc, wc, c,
                                    _ptr_array(dheads), _p(dscale), _s()), "dvq_grain_merge_bwd")
    return dheads, dscale


# ---------------------------------------------------------------------------------------------
# StackGPT building blocks
# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps=1e-5, want_stats=True):
    rows, c = x2d.shape
    y = torch.empty_like(x2d)
    mr = torch.empty(rows, 2, dtype=torch.float32, device=x2d.device) if want_stats else None
    check(lib().dvq_layernorm_fwd(_p(x2d), dt(x2d), rows, c, eps, _p(gamma), _p(beta), _p(y), _p(mr), _s()), "dvq_layernorm_fwd")
    return y, mr


def layernorm_bwd(x2d, dy, mr, gamma, dgamma, dbeta, dres=None, drop=None):
    """dx (+ dres: the gradient of the residual stream that by-passes the normalisation, added in the same pass).
    drop = (p, seed): also returns dropout(dx, p, seed) -- the same decisions as `dropout(dx, p, seed)` -- as a second tensor"""
    rows, c = x2d.shape
    dx = torch.empty_like(x2d)
    ensure_workspace(x2d.device)          # per-workgroup dgamma / dbeta partials + fold kernel instead of a million atomics
    if drop is not None and drop[0] > 0.0:
        dxd = torch.empty_like(x2d)
        check(lib().dvq_layernorm_bwd_res_drop(_p(x2d), _p(dy), _p(dres), dt(x2d), rows, c, _p(mr), _p(gamma), _p(dx), _p(dgamma), _p(dbeta),
                                               _p(dxd), float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF, _s()), "dvq_layernorm_bwd_res_drop")
        return dx, dxd
    check(lib().dvq_layernorm_bwd_res(_p(x2d), _p(dy), _p(dres), dt(x2d), rows, c, _p(mr), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), _s()),
          "dvq_layernorm_bwd_res")
    return dx if drop is None else (dx, None)


def gelu(x):
    y = torch.empty_like(x)
    check(lib().dvq_gelu(_p(x), dt(x), x.numel(), _p(y), _s()), "dvq_gelu")
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().dvq_gelu_bwd(_p(x), _p(dy), dt(x), x.numel(), _p(dx), _s()), "dvq_gelu_bwd")
    return dx


def softmax_causal_(s, rows, length, tq, offset, scale):
    """in place: s <- softmax(scale * s) with the causal mask (row r sees columns <= r % tq + offset)"""
    check(lib().dvq_softmax_causal(_p(s), dt(s), rows, length, tq, offset, scale, _p(s), _s()), "dvq_softmax_causal")
    return s


def embed_gather(idx, table, out, t0, accumulate, bstride=None):
    """out[b, t0:t0+len] (+)= table[idx[b]] ; idx int64 [B,len] (or [len] with bstride=0)"""
    b, ttot, c = out.shape
    ln = idx.shape[-1]
    bs = ln if bstride is None else bstride
    check(lib().dvq_embed_gather(_p(idx), bs, _p(table), dt(out), b, ln, ttot, t0, c, int(accumulate), _p(out), _s()),
          "dvq_embed_gather")


def embed_scatter_add(idx, dout, dtable, t0, padding_idx=-1, bstride=None):
    b, ttot, c = dout.shape
    ln = idx.shape[-1]
    bs = ln if bstride is None else bstride
    check(lib().dvq_embed_scatter_add(_p(idx), bs, _p(dout), dt(dout), b, ln, ttot, t0, c, padding_idx, dtable.shape[0], _p(dtable), _s()),
          "dvq_embed_scatter_add")


def cross_entropy(logits2d, v, target, ignore_index, loss_sum, count, gscale=None, want_grad=False):
    rows, ldl = logits2d.shape
    dl = torch.empty_like(logits2d) if want_grad else None
    check(lib().dvq_cross_entropy(_p(logits2d), dt(logits2d), rows, v, ldl, _p(target), ignore_index, _p(loss_sum), _p(count),
                                  _p(gscale), _p(dl), _s()), "dvq_cross_entropy")
    return dl


def dropout_add(x, a, p, seed):
    """x + dropout(a) in one pass (p = 0: x + a); same decisions as dropout(a, p, seed)"""
    y = torch.empty_like(x)
    check(lib().dvq_dropout_add(_p(x), _p(a), dt(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(y), _s()), "dvq_dropout_add")
    return y


def dropout(x, p, seed):
    y = torch.empty_like(x)
    check(lib().dvq_dropout(_p(x), dt(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(y), _s()), "dvq_dropout")
    return y


def attn_causal_ok(x, n_head, b, t):
    """eligibility of the fused attention kernels (include/dvq_hip.h: bf16, head size 64 / 128, T % 8 == 0, index range)"""
    c = x.shape[-1]
    return (x.dtype == torch.bfloat16 and c in (n_head * 64, n_head * 128) and t % 8 == 0 and b * n_head <= 65535 and b * n_head * t * t < (1 << 32)
            and os.environ.get("DVQ_NO_FUSED_ATTN", "0") != "1")


def _attn_scratch(q, b, t, n_head, backward):
    nbytes = lib().dvq_attn_causal_scratch_bytes(b, t, n_head, q.shape[-1] // n_head, int(backward))
    return torch.empty(nbytes, dtype=torch.uint8, device=q.device)


def attn_causal_drop_mask(q, b, t, n_head):
    """buffer for the forward's dropout keep decisions (1 bit per score of the causal tiles; csrc/attention.hip: drop_tile)"""
    return torch.empty(lib().dvq_attn_causal_mask_bytes(b, t, n_head) // 8, dtype=torch.int64, device=q.device)


def attn_causal_fwd(q, k, v, b, t, n_head, scale, p_drop=0.0, seed=0, drop_mask=None):
    """q, k, v [B*T, C] -> (out [B*T, C], lse fp32 [B, n_head, T]); drop_mask (attn_causal_drop_mask): receives the keep decisions"""
    out = torch.empty_like(q)
    lse = torch.empty(b, n_head, t, dtype=torch.float32, device=q.device)
    scratch = _attn_scratch(q, b, t, n_head, False)
    c = q.shape[-1]
    _timed("attn_causal_fwd", 2 * b * t * t * c, 4 * q.numel() * q.element_size(), lambda: check(
        lib().dvq_attn_causal_fwd(_p(q), _p(k), _p(v), dt(q), b, t, n_head, c // n_head, float(scale), float(p_drop),
                                  int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out), _p(lse), _p(scratch), _p(drop_mask), _s()),
        "dvq_attn_causal_fwd"))
    return out, lse


def attn_causal_bwd(q, k, v, out, dout, lse, b, t, n_head, scale, p_drop=0.0, seed=0, drop_mask=None):
    """drop_mask: the buffer the forward filled with the same (p_drop, seed); None: the decisions are hashed again"""
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    scratch = _attn_scratch(q, b, t, n_head, True)
    c = q.shape[-1]
    _timed("attn_causal_bwd", 5 * b * t * t * c, 8 * q.numel() * q.element_size(), lambda: check(
        lib().dvq_attn_causal_bwd(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), dt(q), b, t, n_head, c // n_head, float(scale),
                                  float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(dq), _p(dk), _p(dv), _p(scratch), _p(drop_mask), _s()),
        "dvq_attn_causal_bwd"))
    return dq, dk, dv


def attn_causal_fwd_fused(qkv, cols, b, t, n_head, scale, p_drop=0.0, seed=0, drop_mask=None):
    """q, k, v as column blocks of ONE projection output qkv [B*T, 3 C] (cols = their first columns, order (q, k, v)): no copies, the
    kernels take the row pitch (dvq_attn_causal_fwd_ld; head size 128) -> (out [B*T, C], lse)"""
    m, ld = qkv.shape
    c = ld // 3
    assert qkv.is_contiguous() and ld == 3 * c and c == n_head * 128
    flat = qkv.view(-1)
    q, k, v = (flat[o:] for o in cols)                 # (pointer offsets: the kernels walk rows with the pitch ld)
    out = torch.empty(m, c, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b, n_head, t, dtype=torch.float32, device=qkv.device)
    _timed("attn_causal_fwd", 2 * b * t * t * c, 4 * m * c * qkv.element_size(), lambda: check(
        lib().dvq_attn_causal_fwd_ld(_p(q), _p(k), _p(v), ld, dt(qkv), b, t, n_head, c // n_head, float(scale), float(p_drop),
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out), _p(lse), _p(drop_mask), _s()),
        "dvq_attn_causal_fwd_ld"))
    return out, lse


def attn_causal_bwd_fused(qkv, cols, out, dout, lse, b, t, n_head, scale, p_drop=0.0, seed=0, drop_mask=None):
    """-> dqkv [B*T, 3 C]: the three gradients as column blocks in the layout of qkv (one input-gradient GEMM consumes them)"""
    m, ld = qkv.shape
    c = ld // 3
    flat = qkv.view(-1)
    q, k, v = (flat[o:] for o in cols)
    dqkv = torch.empty_like(qkv)
    dflat = dqkv.view(-1)
    dq, dk, dv = (dflat[o:] for o in cols)
    scratch = _attn_scratch(out, b, t, n_head, True)
    _timed("attn_causal_bwd", 5 * b * t * t * c, 8 * m * c * qkv.element_size(), lambda: check(
        lib().dvq_attn_causal_bwd_ld(_p(q), _p(k), _p(v), ld, _p(out), _p(dout), _p(lse), dt(qkv), b, t, n_head, c // n_head, float(scale),
                                     float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(dq), _p(dk), _p(dv), _p(scratch), _p(drop_mask),
                                     _s()),
        "dvq_attn_causal_bwd_ld"))
    return dqkv


def attn_full_ok(q, t):
    """eligibility of the fused single-head full attention (AttnBlock): bf16, C = 256, T % 32 == 0"""
    return (q.dtype == torch.bfloat16 and q.shape[-1] == 256 and t % 32 == 0 and q.shape[0] // max(1, t) <= 65535
            and os.environ.get("DVQ_NO_FUSED_ATTNBLOCK", "0") != "1")


def attn_full_fwd(q, k, v, b, t, scale):
    """q, k, v [B*T, C] -> (out [B*T, C], lse fp32 [B, T]); softmax over ALL keys of the image (model.py:168-192)"""
    c = q.shape[-1]
    out = torch.empty_like(q)
    lse = torch.empty(b, t, dtype=torch.float32, device=q.device)
    scratch = torch.empty(lib().dvq_attn_full_scratch_bytes(b, t, c, 0), dtype=torch.uint8, device=q.device)
    _timed("attn_full_fwd", 4 * b * t * t * c, 4 * q.numel() * q.element_size(), lambda: check(
        lib().dvq_attn_full_fwd(_p(q), _p(k), _p(v), dt(q), b, t, c, float(scale), _p(out), _p(lse), _p(scratch), _s()),
        "dvq_attn_full_fwd"))
    return out, lse


def attn_full_bwd(q, k, v, out, dout, lse, b, t, scale):
    c = q.shape[-1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    scratch = torch.empty(lib().dvq_attn_full_scratch_bytes(b, t, c, 1), dtype=torch.uint8, device=q.device)
    _timed("attn_full_bwd", 10 * b * t * t * c, 8 * q.numel() * q.element_size(), lambda: check(
        lib().dvq_attn_full_bwd(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), dt(q), b, t, c, float(scale), _p(dq), _p(dk), _p(dv),
                                _p(scratch), _s()), "dvq_attn_full_bwd"))
    return dq, dk, dv


def decode_stack_scratch(b, c, f, device):
    """zeroed scratch of dvq_decode_stack (activations between its phases + the barrier counters it re-arms itself)"""
    return torch.zeros(lib().dvq_decode_stack_scratch_bytes(b, c, f), dtype=torch.uint8, device=device)


def decode_stack_status(scratch, b, c, f, reset=True):
    """synchronising read-back of the kernel's error word: raises RuntimeError when a device-wide barrier timed out (and re-arms
    the counters so that later launches are not poisoned)"""
    check(lib().dvq_decode_stack_status(_p(scratch), b, c, f, int(reset), _s()), "dvq_decode_stack_status")


def decode_stack(table_dev, n_layers, x, n_head, f, tmax, t_dev, eps, scratch, n_workgroups=0, table_host=None):
    """one token step of all blocks of a transformer (one persistent kernel, or five launches per block); x [B, C] bf16 is updated in
    place.  table_host: the ctypes array the device table was made from (kept alive by the caller)"""
    b, c = x.shape
    host = None if table_host is None else C.cast(table_host, C.c_void_p)
    check(lib().dvq_decode_stack(_p(table_dev), n_layers, b, c, n_head, f, tmax, _p(t_dev), float(eps), _p(x), _p(scratch),
                                 int(n_workgroups), host, _s()), "dvq_decode_stack")
    return x


def attn_decode_dev(q, k_new, v_new, kcache, vcache, n_head, t_dev, scale):
    """device-indexed form (graph replay): append (k_new, v_new) at cache row t_dev[0], attend over rows [0, t]"""
    b, c = q.shape
    out = torch.empty_like(q)
    check(lib().dvq_attn_decode_dev(_p(q), _p(k_new), _p(v_new), _p(kcache), _p(vcache), dt(q), b, n_head, c // n_head, _p(t_dev),
                                    kcache.shape[1], scale, _p(out), _s()), "dvq_attn_decode_dev")
    return out


def rows_dev(x, hidden, t_dev, store):
    """hidden[:, t_dev[0]] <- x (store) or x <- hidden[:, t_dev[0]]; x [B, C], hidden [B, Tmax, C]"""
    b, c = x.shape
    check(lib().dvq_rows_dev(_p(x), _p(hidden), dt(x), b, c, hidden.shape[1], _p(t_dev), int(store), _s()), "dvq_rows_dev")
    return x


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    check(lib().dvq_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, _s()), "dvq_adamw")


def adamw_dev(p, g, m, v, hyper):
    """AdamW step with the hyper-parameters read from device memory (hyper fp32 [8], see include/dvq_hip.h)"""
    check(lib().dvq_adamw_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _s()), "dvq_adamw_dev")


def set_f32x8(dst, values):
    """dst[0:8] <- values (by-value launch arguments: safe to call with the host running many steps ahead)"""
    vals = [float(x) for x in values] + [0.0] * (8 - len(values))
    check(lib().dvq_set_f32x8(_p(dst), *vals, _s()), "dvq_set_f32x8")


def sample_rows(k, n, state):
    """k distinct pseudo-random indices in [0, n) (device-resident RNG state uint64-as-int64 [2]: replayable in a hipGraph)"""
    out = torch.empty(k, dtype=torch.int64, device=state.device)
    check(lib().dvq_sample_rows(_p(out), k, n, _p(state), _s()), "dvq_sample_rows")
    return out


def add_uniform_(x, scale, state):
    """x (fp32, contiguous) += scale * U[0,1) drawn from the device-resident generator state (replay-safe)"""
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(lib().dvq_add_uniform(_p(x), x.numel(), float(scale), _p(state), _s()), "dvq_add_uniform")
    return x


def attn_decode(q, kcache, vcache, n_head, t, scale):
    """q [B,C], caches [B,Tmax,C]; attention of the newest row over the first t cache rows -> [B,C]"""
    b, c = q.shape
    out = torch.empty_like(q)
    check(lib().dvq_attn_decode(_p(q), _p(kcache), _p(vcache), dt(q), b, n_head, c // n_head, t, kcache.shape[1], scale, _p(out), _s()),
          "dvq_attn_decode")
    return out


def sample_constrained(logits2d, temperature, pad_code, state, forbid_idx=None, forbid_from=None, forbid_codes=(), keep_code=-1,
                       late_forbid_code=-1, finished=None, top_k=None, top_p=None, sample=True):
    """one token per row of logits2d [B, V] (V <= 2048) under Dualformer's constraint rules, top-k / top-p, multinomial or top-1:
    ONE launch (csrc/transformer.hip: dvq_sample_constrained).  forbid_idx int64 [B, L] lists columns to mask per row; finished fp32
    [B] (non-zero: the row only keeps pad_code); state = int64 [2] device generator state (key, counter).  -> int64 [B, 1]"""
    b, v = logits2d.shape
    out = torch.empty(b, 1, dtype=torch.int64, device=logits2d.device)
    codes = (C.c_int64 * 4)(*([int(c) for c in forbid_codes] + [-1] * (4 - len(forbid_codes))))
    fi = None
    n_forbid = ld = 0
    if forbid_idx is not None and forbid_idx.shape[1] > 0:
        fi = forbid_idx if forbid_idx.is_contiguous() else forbid_idx.contiguous()
        n_forbid, ld = fi.shape[1], fi.stride(0)
    fin = None
    if finished is not None:
        fin = finished.reshape(-1).to(torch.float32).contiguous()
    assert logits2d.is_cuda and logits2d.stride(1) == 1, "row-major logits (a column slice of a padded matrix is fine)"
    check(lib().dvq_sample_constrained(_praw(logits2d), dt(logits2d), b, v, logits2d.stride(0), float(temperature), _p(fi), n_forbid, ld,
                                       v if forbid_from is None else int(forbid_from), codes, int(keep_code), int(late_forbid_code),
                                       int(pad_code), _p(fin), int(top_k or 0), float(top_p or 0.0), int(bool(sample)), _p(state), _p(out),
                                       _s()), "dvq_sample_constrained")
    return out
