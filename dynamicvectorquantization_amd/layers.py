"""Host-side mirror of the reference's CNN building blocks, running on libdvq_hip kernels.

Mirrors /root/reference/modules/diffusionmodules/model.py:29-192 -- same class names, constructor
arguments, parameter names/shapes (state_dict compatible) and call signatures -- but the compute is
hand-written HIP: every module has

    fwd(x_nhwc, tape)   -> y_nhwc      (tape=None: inference, nothing saved)
    bwd(dy_nhwc, tape)  -> dx_nhwc     (accumulates parameter gradients in place into .grad)

on NHWC tensors of the runtime compute dtype, and ``forward()`` keeps the reference's NCHW call
signature by wrapping fwd/bwd in one ``torch.autograd.Function`` (torch autograd is only the outer
tape; there is no per-op autograd graph and no ATen math on the path).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from . import kernels as K
from . import runtime as rt


class Tape:
    """Per-forward-call storage of what the backward needs (children keyed by name)."""

    def __init__(self):
        self.s = {}
        self.c = {}

    def child(self, name: str) -> "Tape":
        t = self.c.get(name)
        if t is None:
            t = self.c[name] = Tape()
        return t


def _child(tape, name):
    return None if tape is None else tape.child(name)


def to_nhwc(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """NCHW-shaped tensor (any strides) -> contiguous NHWC view/copy of the compute dtype."""
    y = x.permute(0, 2, 3, 1)
    if not y.is_contiguous():
        y = y.contiguous()          # foreign (non channels_last) input: one layout copy at the boundary
    if dtype is not None and y.dtype != dtype:
        y = K.cast(y, dtype)
    return y


def to_nchw(y: torch.Tensor) -> torch.Tensor:
    """contiguous NHWC -> NCHW-shaped channels_last view (no copy)"""
    return y.permute(0, 3, 1, 2)


def _grad_buf(p: nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class HipModule(nn.Module):
    """nn.Module whose forward() is fwd/bwd wrapped in a single autograd node."""

    def forward(self, x, *args, **kwargs):
        if kwargs.pop("_raw", False):
            return self.fwd(x, *args, **kwargs)
        return _ModuleFn.apply(self, torch.is_grad_enabled(), x, *[p for p in self.parameters() if p.requires_grad])

    # NCHW <-> NHWC adapters used by the generic autograd wrapper
    def _fwd_nchw(self, x, tape):
        return to_nchw(self.fwd(to_nhwc(x, rt.compute_dtype()), tape))

    def _bwd_nchw(self, dy, tape, in_dtype):
        dx = self.bwd(to_nhwc(dy, rt.compute_dtype()), tape)
        if dx is None:
            return None
        if dx.dtype != in_dtype:
            dx = K.cast(dx, in_dtype)
        return to_nchw(dx)


class _ModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, want_grad, x, *params):
        ctx.module = module
        ctx.tape = Tape() if want_grad else None    # grad mode is off inside Function.forward: decided by caller
        ctx.in_dtype = x.dtype
        ctx.n_params = len(params)
        with torch.no_grad():
            y = module._fwd_nchw(x, ctx.tape)
        return y

    @staticmethod
    def backward(ctx, dy):
        with torch.no_grad():
            dx = ctx.module._bwd_nchw(dy, ctx.tape, ctx.in_dtype)
        # parameter gradients were accumulated in place into .grad by the kernels
        return (None, None, dx) + (None,) * ctx.n_params


# ---------------------------------------------------------------------------------------------
class _SiluFn(torch.autograd.Function):
    """x * sigmoid(x) on dvq_silu / dvq_silu_bwd (any shape, fp32 or bf16 device tensor; other dtypes are computed in fp32)"""

    @staticmethod
    def _flat8(t, dtype):
        """flat copy in the kernel's dtype, zero-padded to a multiple of 8 elements (one 16-byte vector of bf16)"""
        f = t.contiguous().reshape(-1).to(dtype)
        pad = (-f.numel()) % 8
        return torch.cat([f, f.new_zeros(pad)]) if pad else f

    @staticmethod
    def forward(ctx, x):
        cd = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        xc = _SiluFn._flat8(x, cd)
        ctx.save_for_backward(xc)
        ctx.meta = (x.dtype, x.shape, x.numel())
        return K.silu(xc)[: x.numel()].reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dtype, shape, n = ctx.meta
        return K.silu_bwd(xc, _SiluFn._flat8(dy, xc.dtype))[:n].reshape(shape).to(dtype)


def nonlinearity(x):
    """swish (model.py:29-31) as a standalone differentiable op for reference-side callers; inside the blocks of this package it is
    fused into Normalize.fwd(silu=True) and never launched on its own"""
    return _SiluFn.apply(x)


class Normalize(HipModule):
    """GroupNorm(32, C, eps=1e-6, affine) (model.py:34-35) with optional fused swish."""

    def __init__(self, in_channels, num_groups=32, eps=1e-6):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.fuse_silu = False   # standalone forward(): plain GroupNorm like the reference

    def fwd(self, x, tape, silu=None):
        silu = self.fuse_silu if silu is None else silu
        y, mr = K.gn_forward(x, self.weight, self.bias, self.num_groups, self.eps, silu, stats=getattr(x, "_gn_stats", None))
        if tape is not None:
            tape.s.update(x=x, mr=mr, silu=silu)
        return y

    def prep(self, x, tape):
        """fused path: no normalised tensor is written -- returns the per-(n,c) {scale, shift} table that the consuming
        conv kernel applies (with swish) to its LDS-resident input tile; saves what the GroupNorm backward needs"""
        ss, mr = K.gn_scale_shift(x, self.weight, self.bias, self.num_groups, self.eps, stats=getattr(x, "_gn_stats", None))
        if tape is not None:
            tape.s.update(x=x, mr=mr, silu=True)
        return ss

    def bwd(self, dy, tape, addend=None):
        s = tape.s
        return K.gn_backward(s["x"], dy, s["mr"], self.weight, self.bias, _grad_buf(self.weight), _grad_buf(self.bias),
                             self.num_groups, s["silu"], addend)


class Linear(nn.Module):
    """torch.nn.Linear-compatible parameters (weight [out,in], bias [out], same default init) on the GEMM kernels; rows of
    the activation matrix are NHWC pixels.  Output columns are zero-padded to a multiple of 8 (`out_p`)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.out_p = -(-out_features // 8) * 8
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def _w(self, dtype):
        """compute-dtype copy of the weight (+ zero-padded rows / bias), cached until the parameters change"""
        ep = (rt.param_epoch(self.weight), self.weight.data_ptr(), self.weight._version)
        if dtype == torch.bfloat16 and self.weight.is_cuda and _LINEAR_MULTIPACK:
            # persistent bf16 copies (w and the transposed wt) refreshed for ALL Linear layers of this optimizer group by one launch
            ent = getattr(self, "_lpack", None)
            if ent is None or ent["w"].device != self.weight.device:
                ent = LINEAR_PACKS.register(self)
            if ent["key"] is None:
                # first use of this layer: pack it alone (a group launch here would re-pack every layer registered so far once per
                # NEW layer -- the first forward of a 150-Linear model then did ~11 000 layer packs: 32 ms, profiles/r05_v1_stage2_*)
                LINEAR_PACKS.pack_one(self)
            elif ent["key"] != ep:
                LINEAR_PACKS.repack(self.weight.device, getattr(self.weight, "_dvq_group", 0))
                if ent["key"] != ep:
                    LINEAR_PACKS.pack_one(self)
            if ent["bias_p"] is not None and ent["bias_key"] != ep:
                K.copy_kernel_(ent["bias_p"][: self.out_features], self.bias.detach())
                ent["bias_key"] = ep
            return ent["w"], (ent["bias_p"] if ent["bias_p"] is not None else (self.bias.detach() if self.bias is not None else None))
        hit = getattr(self, "_wcache", None)
        if hit is not None and hit[0] == (ep, dtype):
            return hit[1], hit[2]
        w, b = self._w_uncached(dtype)
        self._wcache = ((ep, dtype), w, b)
        return w, b

    def _wt(self, w):
        """transposed compute-dtype weight for the input-gradient GEMM, cached with (and invalidated by) `_w`'s entry"""
        ent = getattr(self, "_lpack", None)
        if ent is not None and w is ent["w"]:
            return ent["wt"]
        hit = getattr(self, "_wtcache", None)
        if hit is not None and hit[0] is w:
            return hit[1]
        wt = K.transpose(w, 1, self.out_p, self.in_features)
        self._wtcache = (w, wt)
        return wt

    def _wt_ld(self, w):
        """row pitch of `_wt(w)` (out_p unless the layer's transposed copy is a column block of a fused operand)"""
        ent = getattr(self, "_lpack", None)
        return ent.get("wt_ld", self.out_p) if ent is not None and w is ent["w"] else self.out_p

    def _w_uncached(self, dtype):
        w = K.cast(self.weight.detach().contiguous(), dtype)
        b = self.bias.detach() if self.bias is not None else None
        if self.out_p != self.out_features:
            wp = torch.zeros(self.out_p, self.in_features, dtype=dtype, device=w.device)
            K.copy_kernel_(wp[: self.out_features], w)
            w = wp
            if b is not None:
                bp = torch.zeros(self.out_p, dtype=torch.float32, device=w.device)
                K.copy_kernel_(bp[: self.out_features], b)
                b = bp
        return w, b

    def fwd(self, x2d, tape, out=None):
        """x2d [M, in] (compute dtype) -> [M, out_p]; `out`: flat destination allocated by the caller (side-stream launches: the
        result must come from the consumer stream's pool)"""
        m = x2d.shape[0]
        w, b = self._w(x2d.dtype)
        y = K.gemm_nt(x2d, w, m, self.out_p, self.in_features, self.in_features, self.in_features, self.out_p,
                      bias=b, bias_mode=1 if b is not None else 0, out=out)
        if tape is not None:
            tape.s.update(x=x2d, w=w)
        return y.view(m, self.out_p)

    def bwd(self, dy, tape, need_dx=True, addend=None):
        """addend [M, in]: an input gradient this layer's is added to (in the GEMM's epilogue: one rounding, no add kernel)"""
        x2d, w = tape.s["x"], tape.s["w"]
        m = x2d.shape[0]
        # dW[o][i] += sum_m dy[m][o] x[m][i]  straight into the fp32 gradient; db = column sums of dy
        fused_db = self.bias is not None and self.out_p == self.out_features and os.environ.get("DVQ_LINEAR_DB", "fused") == "fused"
        # bias gradient = column sums of dy: taken from the weight-gradient kernel's pass over dy (a separate 17-us reduction per layer
        # otherwise: 2.5 ms of a stage-2 train step)
        def wgrad():
            K.gemm_tn(dy, x2d, m, self.out_features, self.in_features, self.out_p, self.in_features, self.in_features,
                      out=_grad_buf(self.weight), colsum=_grad_buf(self.bias) if fused_db else None)
            if self.bias is not None and not fused_db:
                if self.out_p == self.out_features:
                    K.sum_batch(dy, _grad_buf(self.bias))            # accumulates
                else:
                    db = torch.zeros(self.out_p, dtype=torch.float32, device=dy.device)
                    K.sum_batch(dy, db)
                    _grad_buf(self.bias).add_(db[: self.out_features])

        # the weight gradient has no consumer before the optimizer step / gradient exchange: on the side stream (runtime.side_wgrad)
        # its workgroups fill the partially occupied last round of the input-gradient GEMM that runs beside it (20576-row operands
        # give 324 / 432 tiles for 256 CUs) and its HBM-bound partial fold overlaps MFMA-bound kernels.  DVQ_LINEAR_SIDE=0: main stream
        if need_dx and m >= 1024 and rt.side_wgrad_enabled() and os.environ.get("DVQ_LINEAR_SIDE", "1") != "0":
            wt = self._wt(w)                                                 # made on the main stream, before the fork
            dx = K.gemm_nt(dy, wt, m, self.in_features, self.out_p, self.out_p, self._wt_ld(w), self.in_features, residual=addend)
            rt.run_on_side(wgrad, dy, x2d)      # after the input gradient (forking before it measured the same: 82.3 vs 82.0 ms)
            return dx.view(m, self.in_features)
        wgrad()
        if not need_dx:
            return None
        wt = self._wt(w)                                                     # [in, out_p]
        dx = K.gemm_nt(dy, wt, m, self.in_features, self.out_p, self.out_p, self._wt_ld(w), self.in_features, residual=addend)
        return dx.view(m, self.in_features)


class _LinearPackRegistry:
    """bf16 copies of every Linear weight of the process: w [out_p, in] and wt [in, out_p] in persistent buffers, refreshed after an
    optimizer step by ONE launch per optimizer group (dvq_linear_pack_multi) instead of a cast and a transpose launch per layer --
    291 launches / 4.3 ms of the StackGPT p6c18 train step."""

    def __init__(self):
        self.items = {}       # device -> list of weakref(module)
        self.tables = {}      # (device, group) -> dict(sig, table, n, tiles)

    @staticmethod
    def _key(m):
        return (rt.param_epoch(m.weight), m.weight.data_ptr(), m.weight._version)

    def register(self, mod, w=None, wt=None, wt_ld=0, bias_dst=None):
        """w / wt / bias_dst: caller-owned destinations instead of private buffers -- slices of the row-concatenated operand of several
        projections that share their input (stackgpt.CausalSelfAttention: [Wk; Wq; Wv], its transpose with row pitch wt_ld, and the
        concatenated bias); wt is then a FLAT tensor starting at the layer's first column"""
        import weakref
        dev = mod.weight.device
        known = getattr(mod, "_lpack", None) is not None
        ent = {"w": w if w is not None else torch.zeros(mod.out_p, mod.in_features, dtype=torch.bfloat16, device=dev),
               "wt": wt if wt is not None else torch.zeros(mod.in_features, mod.out_p, dtype=torch.bfloat16, device=dev),
               "wt_ld": int(wt_ld) if wt is not None else mod.out_p, "bias_dst": bias_dst, "key": None, "bias_key": None,
               "bias_p": (torch.zeros(mod.out_p, dtype=torch.float32, device=dev)
                          if mod.bias is not None and mod.out_p != mod.out_features else None)}
        mod._lpack = ent
        if not known:
            self.items.setdefault(dev, []).append(weakref.ref(mod))
        for k in [k for k in self.tables if k[0] == dev]:
            self.tables.pop(k, None)
        return ent

    def _launch(self, mods, cache_key=None):
        from ._lib import LinPackEntry
        sig = tuple((m.weight.data_ptr(), m._lpack["w"].data_ptr()) for m in mods)
        tab = self.tables.get(cache_key) if cache_key is not None else None
        if tab is None or tab["sig"] != sig:
            arr = (LinPackEntry * len(mods))()
            begin = 0
            for i, m in enumerate(mods):
                e = m._lpack
                bd = e.get("bias_dst")
                arr[i] = LinPackEntry(m.weight.data_ptr(), e["w"].data_ptr(), e["wt"].data_ptr(), m.out_features, m.in_features, m.out_p,
                                      begin, e.get("wt_ld", m.out_p), m.bias.data_ptr() if bd is not None else None,
                                      bd.data_ptr() if bd is not None else None)
                begin += -(-m.out_p // 64) * -(-m.in_features // 64)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(mods[0].weight.device)
            tab = {"sig": sig, "table": raw, "n": len(mods), "tiles": begin}
            if cache_key is not None:
                self.tables[cache_key] = tab
        K.linear_pack_multi(tab["table"], tab["n"], tab["tiles"])
        for m in mods:
            m._lpack["key"] = self._key(m)
        return tab

    def repack(self, device, group=0):
        mods = [m for m in (r() for r in self.items.get(device, [])) if m is not None and getattr(m, "_lpack", None) is not None and
                m.weight.device == device and m._lpack["w"].device == device and getattr(m.weight, "_dvq_group", 0) == group and
                m.weight.is_contiguous() and m.weight.dtype == torch.float32]
        if len(mods) >= 2:
            self._launch(mods, (device, group))

    def pack_one(self, mod):
        assert mod.weight.dtype == torch.float32
        if not mod.weight.is_contiguous():
            raise RuntimeError("Linear weight must be contiguous")
        mod._lpack["_one"] = self._launch([mod], None)          # (the table tensor must outlive the launch)


LINEAR_PACKS = _LinearPackRegistry()
_LINEAR_MULTIPACK = os.environ.get("DVQ_LINEAR_MULTIPACK", "1") != "0"


class BatchNorm2d(HipModule):
    """torch.nn.BatchNorm2d(C) (eps 1e-5, momentum 0.1, affine, running statistics) with the LeakyReLU(0.2) that
    follows it in the PatchGAN fused in (modules/discriminator/model.py:44-60).  Training mode normalises with the
    batch statistics: that is GroupNorm with one group per channel over the whole batch, so it runs on the dvq_gn_*
    kernels with the tensor viewed as [1, N*H*W, C].  Eval mode uses the running statistics (per-channel affine)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.act = K.ACT_NONE

    def fwd(self, x, tape, act=None, update_running=True):
        act = self.act if act is None else act
        n, h, w, c = x.shape
        flat = x.view(1, n * h * w, c)
        if self.training:
            stats = K.gn_stats(flat, c)
            y, mr = K.gn_forward(flat, self.weight, self.bias, c, self.eps, act, stats=stats)
            if update_running:
                cnt = float(n * h * w)
                mean = stats[0, :, 0] / cnt
                var = (stats[0, :, 1] / cnt - mean * mean).clamp_(min=0) * (cnt / max(cnt - 1.0, 1.0))
                self.running_mean.mul_(1 - self.momentum).add_(mean.float(), alpha=self.momentum)
                self.running_var.mul_(1 - self.momentum).add_(var.float(), alpha=self.momentum)
                self.num_batches_tracked += 1
        else:
            # eval: the running statistics are turned into the {sum, sum of squares} of a unit count, so the same
            # kernel applies (x - mean) / sqrt(var + eps)
            stats = torch.stack([self.running_mean.double(), (self.running_var + self.running_mean ** 2).double()], dim=1)
            stats = (stats * float(n * h * w)).view(1, c, 2).contiguous()
            y, mr = K.gn_forward(flat, self.weight, self.bias, c, self.eps, act, stats=stats)
        if tape is not None:
            tape.s.update(x=flat, mr=mr, act=act, shape=x.shape)
        return y.view(n, h, w, c)

    def bwd(self, dy, tape, need_dw=True):
        s = tape.s
        c = self.num_features
        if need_dw:
            dg, db = _grad_buf(self.weight), _grad_buf(self.bias)
        else:
            dg, db = torch.zeros_like(self.weight), torch.zeros_like(self.bias)
        dx = K.gn_backward(s["x"], dy.view(s["x"].shape), s["mr"], self.weight, self.bias, dg, db, c, s["act"])
        return dx.view(s["shape"])


class ActNorm(HipModule):
    """utils/utils.py:58-140 (the forward direction the PatchGAN uses with use_actnorm=True, modules/discriminator/model.py:30-33):
    h = scale * (x + loc) per channel, loc / scale initialised from the first training batch (loc = -mean, scale = 1 / (std + 1e-6),
    unbiased std) and trained afterwards; the LeakyReLU that follows is fused in.  Runs on the normalisation kernels with CONSTANT
    statistics (mean = -loc, rstd = 1, gamma = scale): forward = dvq_gn_apply, backward = the reduction (d scale = sum dz (x + loc),
    d loc = scale * sum dz) + dx = scale * dz."""

    EPS = 1e-5

    def __init__(self, num_features, logdet=False, affine=True, allow_reverse_init=False):
        assert affine
        super().__init__()
        if logdet:
            raise NotImplementedError("ActNorm(logdet=True) belongs to the flow models, not to this repository's path")
        self.logdet, self.allow_reverse_init, self.num_features = logdet, allow_reverse_init, num_features
        self.loc = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.scale = nn.Parameter(torch.ones(1, num_features, 1, 1))
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))
        self._inited = False                   # host-side mirror of `initialized` (None = unknown: re-read from the buffer)
        self.act = K.ACT_NONE

    def fwd(self, x, tape, act=None, update_running=True):
        act = self.act if act is None else act
        n, h, w, c = x.shape
        flat = x.view(1, n * h * w, c)
        if self.training and not self.inited():                         # data-dependent initialisation (one host sync, once)
            with torch.no_grad():
                f = x.float().reshape(-1, c)
                self.loc.data.copy_((-f.mean(0)).view(1, c, 1, 1))
                self.scale.data.copy_((1.0 / (f.std(0) + 1e-6)).view(1, c, 1, 1))
                self.initialized.fill_(1)
            self._inited = True
        cnt = float(n * h * w)
        mean = -self.loc.detach().view(c).double()
        stats = (torch.stack([mean, (1.0 - self.EPS) + mean * mean], dim=1) * cnt).view(1, c, 2).contiguous()
        zero = torch.zeros(c, dtype=torch.float32, device=x.device)
        y, mr = K.gn_forward(flat, self.scale.detach().view(c), zero, c, self.EPS, act, stats=stats)
        if tape is not None:
            tape.s.update(x=flat, mr=mr, act=act, shape=x.shape)
        return y.view(n, h, w, c)

    def inited(self) -> bool:
        """host-side copy of the `initialized` buffer: read from the device ONCE (a per-forward .item() is a host sync per layer and
        step, and illegal during stream capture); refreshed when a state_dict is loaded"""
        if self._inited is None:
            self._inited = bool(int(self.initialized.item()) != 0)
        return self._inited

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._inited = None

    def bwd(self, dy, tape, need_dw=True):
        s = tape.s
        c = self.num_features
        dscale = torch.zeros(c, dtype=torch.float32, device=dy.device)
        dsum = torch.zeros(c, dtype=torch.float32, device=dy.device)
        zero = torch.zeros(c, dtype=torch.float32, device=dy.device)
        dx = K.gn_backward(s["x"], dy.view(s["x"].shape), s["mr"], self.scale.detach().view(c), zero, dscale, dsum, c, s["act"],
                           fixed_stats=True)
        if need_dw:
            _grad_buf(self.scale).view(c).add_(dscale)
            _grad_buf(self.loc).view(c).add_(dsum * self.scale.detach().view(c))
        return dx.view(s["shape"])


_SIDE_AFTER_DGRAD = os.environ.get("DVQ_SIDE_AFTER_DGRAD", "1") != "0"


class Conv2d(HipModule):
    """torch.nn.Conv2d-compatible parameters ([Cout,Cin,KH,KW] + bias, same default init) driving the
    implicit-GEMM kernels.  `asym_pad` reproduces Downsample's F.pad(0,1,0,1); `upsample` reads the
    input through nearest x2 (Upsample) without materialising it."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, asym_pad=False,
                 upsample=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.asym_pad, self.upsample = asym_pad, upsample
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packs = {}

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_channels * self.kernel_size ** 2)
            nn.init.uniform_(self.bias, -bound, bound)

    def _padded(self, dtype):
        v = K.vec(dtype)
        return -(-self.in_channels // v) * v, -(-self.out_channels // v) * v

    def _alloc_pack(self, dtype):
        """persistent packed buffers (stable device pointers: the multi-tensor pack table refers to them)"""
        cin_p, cout_p = self._padded(dtype)
        k, dev = self.kernel_size, self.weight.device
        w = torch.zeros(cout_p, k, k, cin_p, dtype=dtype, device=dev)          # rows >= Cout stay zero
        wt = torch.zeros(cin_p, k, k, cout_p, dtype=dtype, device=dev)                # rows >= Cin stay zero (dgrad reads cin_p rows)
        bias = None
        if self.bias is not None and cout_p != self.out_channels:
            bias = torch.zeros(cout_p, dtype=torch.float32, device=dev)
        ent = {"epoch": -1, "w": w, "wt": wt, "bias": bias, "cin_p": cin_p, "cout_p": cout_p, "master": self.weight.data_ptr()}
        self._packs[dtype] = ent
        PACKS.register(self, dtype)
        return ent

    def _pack_key(self):
        """what the packed copies were made from: the runtime's parameter epochs (optimizer steps, whole-model
        load_state_dict) AND the tensors' own version counters (submodule load_state_dict, weights_init-style or manual
        in-place edits after a first forward)"""
        return (rt.param_epoch(self.weight), self.weight._version, None if self.bias is None else self.bias._version)

    def packed(self, dtype):
        ent = self._packs.get(dtype)
        if ent is None or ent["w"].device != self.weight.device or ent["master"] != self.weight.data_ptr():
            ent = self._alloc_pack(dtype)
        ep = self._pack_key()
        if ent["epoch"] != ep:
            if ent["epoch"] != -1:
                # ONE launch refreshes every registered conv of this dtype / device / optimizer group
                PACKS.repack(dtype, self.weight.device, getattr(self.weight, "_dvq_group", 0))
            if ent["epoch"] != ep:                       # not covered by the table yet (first use)
                K.pack_weight_into(self.weight.detach(), ent["cin_p"], ent["cout_p"], dtype, ent["w"], ent["wt"])
                ent["epoch"] = ep
        if ent["bias"] is not None and ent.get("bias_epoch") != ep:
            # zero-padded bias copy (Cout not a multiple of the vector width): tracked on its own -- the multi-tensor
            # re-pack triggered by ANOTHER conv refreshes this module's weights but not this small copy
            K.copy_kernel_(ent["bias"][: self.out_channels], self.bias.detach())
            ent["bias_epoch"] = ep
        bias = ent["bias"] if ent["bias"] is not None else (self.bias.detach() if self.bias is not None else None)
        return ent["w"], ent["wt"], bias

    def _desc(self, x):
        n, h, w, _ = x.shape
        if self.upsample:
            h, w = 2 * h, 2 * w
        k, s = self.kernel_size, self.stride
        if self.asym_pad:
            pt = pl = 0
            oh, ow = (h + 1 - k) // s + 1, (w + 1 - k) // s + 1
        else:
            pt = pl = self.padding
            oh, ow = (h + 2 * self.padding - k) // s + 1, (w + 2 * self.padding - k) // s + 1
        cin_p, cout_p = self._padded(x.dtype)
        assert x.shape[-1] == cin_p, f"expected {cin_p} (padded) input channels, got {x.shape[-1]}"
        return K.conv_desc(n, h, w, cin_p, cout_p, k, k, s, pt, pl, oh, ow, self.upsample, x.dtype, rt.impl())

    def fused_ok(self, x) -> bool:
        """True if this call runs on the halo kernels, which can apply GroupNorm+swish to their input tile and emit
        the GroupNorm statistics of their output"""
        return x.dtype == torch.bfloat16 and self.kernel_size == 3 and K.conv_fused_ok(self._desc(x))

    def fwd(self, x, tape, residual=None, gn_ss=None, want_stats=False, act=K.ACT_NONE):
        w, _, bias = self.packed(x.dtype)
        d = self._desc(x)
        if act != K.ACT_NONE:           # ReLU / LeakyReLU fused into the conv epilogue (VGG16, PatchGAN)
            y = K.conv2d_fwd(d, x, w, bias, act=act)
            if tape is not None:
                tape.s.update(x=x, d=d, gn_ss=None)
            return y
        stats = None
        if want_stats and self.out_channels % 32 == 0 and 128 % (self.out_channels // 32) == 0 and self.fused_ok(x):
            stats = K.zeros_small((d.N, 32, 2), torch.float64, x.device)
        y = K.conv2d_fwd(d, x, w, bias, residual, gn_ss=gn_ss, out_stats=stats, out_groups=32 if stats is not None else 0)
        if stats is not None:
            y._gn_stats = stats            # consumed by the next Normalize (saves its statistics pass)
        if tape is not None:
            tape.s.update(x=x, d=d, gn_ss=gn_ss)
        return y

    def bwd(self, dy, tape, need_dx=True, need_dw=True, mask=None, mask_act=K.ACT_NONE):
        """mask / mask_act: output and kind of the activation that produced this conv's input (its gate is applied to dx);
        need_dw=False: frozen parameters (LPIPS' VGG16, the discriminator during the generator update)"""
        x, d = tape.s["x"], tape.s["d"]
        side = need_dw and need_dx and rt.side_wgrad_enabled()
        dx = None
        if need_dx and side and _SIDE_AFTER_DGRAD:
            # the input gradient FIRST: the side stream then waits for it, so the weight gradient (matrix pipes) runs beside what
            # follows the input gradient on this stream -- the HBM-bound GroupNorm backward -- instead of beside the input
            # gradient itself, which wants the same matrix pipes
            _, wt, _ = self.packed(x.dtype)
            dx = K.conv2d_dgrad(d, dy, wt, mask=mask, mask_act=mask_act)
        if need_dw:
            db = _grad_buf(self.bias) if self.bias is not None else None
            # weight / bias gradients are accumulated by the kernel straight into the reference-layout .grad buffers
            gw, gss = _grad_buf(self.weight), tape.s.get("gn_ss")
            if side:
                # on the side stream: nothing reads this gradient before the optimizer step / its bucket's exchange
                rt.run_on_side(lambda: K.conv2d_wgrad_oihw(d, x, dy, self.in_channels, self.out_channels, gw, db, gn_ss=gss), x, dy)
            else:
                K.conv2d_wgrad_oihw(d, x, dy, self.in_channels, self.out_channels, gw, db, gn_ss=gss)
        if not need_dx:
            return None
        if dx is not None:
            return dx
        _, wt, _ = self.packed(x.dtype)
        return K.conv2d_dgrad(d, dy, wt, mask=mask, mask_act=mask_act)


def conv1x1_bwd_accumulate(conv, dy, tape, acc):
    """conv.bwd(dy, tape) + acc for a 1 x 1 / stride 1 convolution in ONE kernel: its input gradient is the forward 1 x 1 convolution
    of dy with the transposed weight, whose epilogue adds a residual -- the separate dvq_add pass over the gradient (3 tensors of HBM
    traffic: as long as the GEMM itself at 256 channels) disappears.  The weight gradient is issued as in Conv2d.bwd."""
    x, d = tape.s["x"], tape.s["d"]
    ok = (conv.kernel_size == 1 and conv.stride == 1 and conv.padding == 0 and not conv.upsample and not conv.asym_pad and
          dy.dtype == torch.bfloat16 and rt.impl() == 0 and acc is not None and d.Cin == conv.in_channels and d.Cout == conv.out_channels)
    if not ok:
        dx = conv.bwd(dy, tape)
        return dx if acc is None else K.add(dx, acc)
    _, wt, _ = conv.packed(x.dtype)
    dt_ = K.conv_desc(d.N, d.H, d.W, d.Cout, d.Cin, 1, 1, 1, 0, 0, d.H, d.W, False, dy.dtype, d.impl)
    dx = K.conv2d_fwd(dt_, dy, wt, None, acc)
    db = _grad_buf(conv.bias) if conv.bias is not None else None
    gw = _grad_buf(conv.weight)
    if rt.side_wgrad_enabled():
        rt.run_on_side(lambda: K.conv2d_wgrad_oihw(d, x, dy, conv.in_channels, conv.out_channels, gw, db), x, dy)
    else:
        K.conv2d_wgrad_oihw(d, x, dy, conv.in_channels, conv.out_channels, gw, db)
    return dx


class _PackRegistry:
    """All Conv2d packed-weight buffers of the process; `repack` refreshes every one of them with ONE kernel
    launch (dvq_pack_weights_multi) after an optimizer step instead of one launch per layer."""

    def __init__(self):
        self.items = {}       # (dtype, device) -> list of (weakref(module))
        self.tables = {}      # (dtype, device) -> dict(sig, table tensor, n, total)

    def register(self, mod, dtype):
        import weakref
        key = (dtype, mod.weight.device)
        self.items.setdefault(key, []).append(weakref.ref(mod))
        for k in [k for k in self.tables if k[:2] == key]:
            self.tables.pop(k, None)

    def repack(self, dtype, device, group=0):
        import ctypes
        from ._lib import PackEntry
        key = (dtype, device)
        mods = [m for m in (r() for r in self.items.get(key, [])) if m is not None and dtype in m._packs and
                getattr(m.weight, "_dvq_group", 0) == group]
        key = (dtype, device, group)
        live = [m for m in mods if m._packs[dtype]["master"] == m.weight.data_ptr() and m._packs[dtype]["w"].device == device]
        if len(live) < 2:
            return
        sig = tuple((m.weight.data_ptr(), m._packs[dtype]["w"].data_ptr(), m.weight.stride()) for m in live)
        tab = self.tables.get(key)
        if tab is None or tab["sig"] != sig:
            arr = (PackEntry * len(live))()
            begin = 0
            for i, m in enumerate(live):
                e = m._packs[dtype]
                k2 = m.kernel_size * m.kernel_size
                arr[i] = PackEntry(m.weight.data_ptr(), e["w"].data_ptr(), e["wt"].data_ptr(), m.out_channels, m.in_channels,
                                   k2, e["cin_p"], e["cout_p"], begin, K.dt(dtype) | (256 if K.is_ohwi(m.weight) else 0))
                begin += m.out_channels * k2 * e["cin_p"] + m.in_channels * k2 * e["cout_p"]
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            tab = {"sig": sig, "table": raw, "n": len(live), "total": begin}
            self.tables[key] = tab
        K.pack_weights_multi(tab["table"], tab["n"], tab["total"])
        for m in live:
            m._packs[dtype]["epoch"] = m._pack_key()


PACKS = _PackRegistry()


class Upsample(HipModule):
    """nearest x2 (+ 3x3 conv: the upsampled tensor is then never materialised) (model.py:38-53)"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = Conv2d(in_channels, in_channels, 3, stride=1, padding=1, upsample=True)

    def fwd(self, x, tape):
        if not self.with_conv:
            return K.upsample_nearest2x(x)
        return self.conv.fwd(x, _child(tape, "conv"), want_stats=True)

    def bwd(self, dy, tape):
        if not self.with_conv:
            return K.upsample_nearest2x_bwd(dy)
        return self.conv.bwd(dy, tape.child("conv"))


class Downsample(HipModule):
    """pad (0,1,0,1) + 3x3 stride-2 conv (the padding is folded into the gather), or avg_pool2d(2, 2) (model.py:56-75)"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = Conv2d(in_channels, in_channels, 3, stride=2, padding=0, asym_pad=True)

    def fwd(self, x, tape):
        if not self.with_conv:
            n, h, w, c = x.shape
            y = torch.empty(n, h // 2, w // 2, c, dtype=x.dtype, device=x.device)
            K.avgpool_slice(x[:, : 2 * (h // 2), : 2 * (w // 2)].contiguous() if (h | w) & 1 else x, 2, y, 0)
            if tape is not None:
                tape.s["in_hw"] = (h, w)
            return y
        return self.conv.fwd(x, _child(tape, "conv"))

    def bwd(self, dy, tape):
        if not self.with_conv:
            dx = K.avgpool_slice_bwd(dy, 0, dy.shape[-1], 2)
            h, w = tape.s["in_hw"]
            if (h | w) & 1:                           # odd sizes: avg_pool2d drops the last row / column
                dx = torch.nn.functional.pad(dx, (0, 0, 0, w - dx.shape[2], 0, h - dx.shape[1]))
            return dx
        return self.conv.bwd(dy, tape.child("conv"))


def norm_swish_conv(norm, conv, x, tape, nname, cname, residual=None):
    """GroupNorm -> swish -> conv3x3 (model.py:119-129).  On shapes the halo kernel takes, the normalised activation
    is never materialised: the conv applies scale/shift + swish to its LDS input tile and also emits the GroupNorm
    statistics of its output for whoever normalises it next."""
    # with a backward to come the weight-gradient kernel would have to re-apply the transform (measured: as expensive as the
    # gn_apply pass it saves), so the prologue fusion is used for tape-less forwards only: inference, and the discriminator
    # step's second autoencoder forward (DVQ_FUSE_GN=1 forces it everywhere)
    if (rt.fuse_gn_prologue() or (tape is None and rt.fuse_gn_inference())) and conv.fused_ok(x):
        ss = norm.prep(x, _child(tape, nname))
        return conv.fwd(x, _child(tape, cname), residual=residual, gn_ss=ss, want_stats=True)
    a = norm.fwd(x, _child(tape, nname), silu=True)
    return conv.fwd(a, _child(tape, cname), residual=residual, want_stats=True)


class ResnetBlock(HipModule):
    """GN-swish-conv3x3-GN-swish-conv3x3 + (1x1) shortcut (model.py:78-137); temb is always None."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if temb_channels > 0:
            raise NotImplementedError("temb_channels > 0 is unused by the DQ-VAE (temb is None)")
        self.dropout = nn.Dropout(dropout)        # (model.py:97; applied between norm2 + swish and conv2, :127)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:                         # model.py:103-108: a 3x3 shortcut under the reference's parameter name
                self.conv_shortcut = Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, 1, 1, 0)

    def _shortcut(self):
        return self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut

    def forward(self, x, temb=None, **kw):
        assert temb is None
        return super().forward(x, **kw)

    def fwd(self, x, tape, temb=None):
        h1 = norm_swish_conv(self.norm1, self.conv1, x, tape, "norm1", "conv1")
        sc = self._shortcut().fwd(x, _child(tape, "nin")) if self.in_channels != self.out_channels else x
        if self.training and self.dropout.p > 0.0:
            # dropout between the activation and conv2: the activation is materialised (no prologue fusion); the keep decisions are
            # a hash of (seed, element index) -- the backward re-derives them from the seed (csrc/dvq_common.h: dvq_hash32)
            a = self.norm2.fwd(h1, _child(tape, "norm2"), silu=True)
            seed = rt.next_dropout_seed()             # host-side: a model with dropout > 0 is not step-recorded (dqvae.graph_signature)
            if tape is not None:
                tape.s["drop"] = (float(self.dropout.p), seed)
            a = K.dropout(a, float(self.dropout.p), seed)
            return self.conv2.fwd(a, _child(tape, "conv2"), residual=sc, want_stats=True)
        return norm_swish_conv(self.norm2, self.conv2, h1, tape, "norm2", "conv2", residual=sc)

    def bwd(self, dy, tape):
        d = self.conv2.bwd(dy, tape.child("conv2"))
        if "drop" in tape.s:
            d = K.dropout(d, *tape.s["drop"])
        d = self.norm2.bwd(d, tape.child("norm2"))
        d = self.conv1.bwd(d, tape.child("conv1"))
        sc = self._shortcut().bwd(dy, tape.child("nin")) if self.in_channels != self.out_channels else dy
        return self.norm1.bwd(d, tape.child("norm1"), addend=sc)      # skip-path gradient added in the same pass


class AttnBlock(HipModule):
    """single-head spatial self-attention with 1x1-conv q/k/v/proj (model.py:140-192)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, 1)
        self.k = Conv2d(in_channels, in_channels, 1)
        self.v = Conv2d(in_channels, in_channels, 1)
        self.proj_out = Conv2d(in_channels, in_channels, 1)

    def fwd(self, x, tape):
        b, h, w, c = x.shape
        n = h * w
        hn = self.norm.fwd(x, _child(tape, "norm"), silu=False)
        q = self.q.fwd(hn, _child(tape, "q"))
        k = self.k.fwd(hn, _child(tape, "k"))
        v = self.v.fwd(hn, _child(tape, "v"))
        impl = rt.impl()
        if impl == 0 and K.attn_full_ok(q.view(b * n, c), n):
            # fused single-head attention (flash recurrence, head size C = 256): the [B,N,N] scores never reach HBM
            o, lse = K.attn_full_fwd(q.view(b * n, c), k.view(b * n, c), v.view(b * n, c), b, n, float(int(c) ** (-0.5)))
            y = self.proj_out.fwd(o.view(b, h, w, c), _child(tape, "proj"), residual=x)
            if tape is not None:
                tape.s.update(q=q, k=k, v=v, o=o, lse=lse, p=None, shape=(b, h, w, c))
            return y
        s = K.gemm_nt(q, k, n, n, c, c, c, n, batch=b, sa=n * c, sb=n * c, sc=n * n, impl=impl)       # q k^T
        p = K.softmax_rows(s, b * n, n, float(int(c) ** (-0.5)))
        vt = K.transpose(v, b, n, c)                                                                    # [b,c,n]
        o = K.gemm_nt(p, vt, n, c, n, n, n, c, batch=b, sa=n * n, sb=c * n, sc=n * c, impl=impl)       # p v
        o = o.view(b, h, w, c)
        y = self.proj_out.fwd(o, _child(tape, "proj"), residual=x)
        if tape is not None:
            tape.s.update(q=q, k=k, v=v, p=p, shape=(b, h, w, c))
        return y

    def bwd(self, dy, tape):
        st = tape.s
        b, h, w, c = st["shape"]
        n = h * w
        q, k, v, p = st["q"], st["k"], st["v"], st["p"]
        impl = rt.impl()
        do = self.proj_out.bwd(dy, tape.child("proj"))
        if p is None and os.environ.get("DVQ_ATTNBLOCK_BWD", "flash") == "gemm":
            # DVQ_ATTNBLOCK_BWD=gemm: fused forward, backward on the pipelined batched GEMM kernels -- the probabilities are recomputed
            # (q k^T, row softmax) into scratch that lives only inside this call.  This was the default while the flash-style backward at
            # head size 256 ran at 175 TFLOP/s (first-generation kernels, 0.98 ms per call at B 64, T 1024, against 0.68 ms here); the
            # round-6 kernels (csrc/attention2.hip) take 0.45 ms, so the fused backward below is the default again.
            s = K.gemm_nt(q, k, n, n, c, c, c, n, batch=b, sa=n * c, sb=n * c, sc=n * n, impl=impl)
            p = K.softmax_rows(s, b * n, n, float(int(c) ** (-0.5)))
            del s
        if p is None:                                   # fused forward: fused backward (dQ, then dV and dK kernels)
            dq, dk, dv = K.attn_full_bwd(q.view(b * n, c), k.view(b * n, c), v.view(b * n, c), st["o"], do.view(b * n, c), st["lse"],
                                         b, n, float(int(c) ** (-0.5)))
            dh = self.q.bwd(dq.view(b, h, w, c), tape.child("q"))
            dh = conv1x1_bwd_accumulate(self.k, dk.view(b, h, w, c), tape.child("k"), dh)
            dh = conv1x1_bwd_accumulate(self.v, dv.view(b, h, w, c), tape.child("v"), dh)
            return self.norm.bwd(dh, tape.child("norm"), addend=dy)
        dp = K.gemm_nt(do, v, n, n, c, c, c, n, batch=b, sa=n * c, sb=n * c, sc=n * n, impl=impl)      # dO v^T
        dv32 = K.gemm_tn(p, do, n, n, c, n, c, c, batch=b, sa=n * n, sb=n * c, sc=n * c, impl=impl)    # p^T dO
        ds = K.softmax_rows_bwd(p, dp, b * n, n, float(int(c) ** (-0.5)))
        kt = K.transpose(k, b, n, c)
        dq = K.gemm_nt(ds, kt, n, c, n, n, n, c, batch=b, sa=n * n, sb=c * n, sc=n * c, impl=impl)     # dS k
        dk32 = K.gemm_tn(ds, q, n, n, c, n, c, c, batch=b, sa=n * n, sb=n * c, sc=n * c, impl=impl)    # dS^T q
        dt_ = q.dtype
        dq = dq.view(b, h, w, c)
        dk = K.cast(dk32.view(b, h, w, c), dt_)
        dv = K.cast(dv32.view(b, h, w, c), dt_)
        dh = self.q.bwd(dq, tape.child("q"))
        dh = conv1x1_bwd_accumulate(self.k, dk, tape.child("k"), dh)
        dh = conv1x1_bwd_accumulate(self.v, dv, tape.child("v"), dh)
        return self.norm.bwd(dh, tape.child("norm"), addend=dy)
