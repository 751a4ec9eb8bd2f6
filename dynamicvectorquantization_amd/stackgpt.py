"""StackGPT (stacked position / content transformers of the DQ-Transformer) on libdvq_hip kernels.

Mirrors /root/reference/modules/dynamic_modules/stackgpt.py:7-339: same constructor arguments, parameter / buffer names
(`content_emb.weight`, `pos_emb`, `position_transformer.3.attn.key.weight`, `content_head.1.weight`, ...) and
forward() contract (training: dict of the four losses; otherwise the two logit tensors).

Execution: activations are [B*T, C] row matrices in the runtime compute dtype; T is padded to a multiple of 8 inside (the
padded tail sits behind the causal mask and its targets are the ignore index, so results are unaffected) so every GEMM is
MFMA-aligned.  Per block: LayerNorm kernels, q/k/v/proj/MLP on the GEMM kernels, attention as per-head strided batched
GEMMs (QK^T -> causal softmax kernel -> PV), GELU / dropout kernels.  The whole model is ONE autograd node whose backward
walks the tapes (gradients accumulate in place into .grad / the flat optimizer buffers).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from . import kernels as K
from . import runtime as rt
from .layers import Linear, Tape, _child, _grad_buf


class StackGPTConfig:
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


_next_seed = rt.next_dropout_seed


def _fuse_drop_bwd() -> bool:
    return os.environ.get("DVQ_FUSE_DROP_BWD", "1") != "0"


def _drop(x, p, training, tape, key):
    if not training or p <= 0.0:
        return x
    seed = _next_seed()
    if tape is not None:
        tape.s[key] = (p, seed)
    return K.dropout(x, p, seed)


def _drop_add(x, a, p, training, tape, key):
    """x + dropout(a) in one pass (the residual add + resid_drop of a block, stackgpt.py:66-69,91-96 of the reference)"""
    if not training or p <= 0.0:
        return K.add(x, a)
    seed = _next_seed()
    if tape is not None:
        tape.s[key] = (p, seed)
    return K.dropout_add(x, a, p, seed)


def _drop_bwd(g, tape, key):
    ps = tape.s.get(key)
    return g if ps is None else K.dropout(g, ps[0], ps[1])


class LayerNorm(nn.Module):
    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))

    def fwd(self, x2d, tape):
        y, mr = K.layernorm_fwd(x2d, self.weight, self.bias, self.eps, want_stats=tape is not None)
        if tape is not None:
            tape.s.update(x=x2d, mr=mr)
        return y

    def bwd(self, dy, tape, dres=None, drop=None):
        """dres: gradient of the residual stream around this normalisation, added to the result in the same pass.
        drop = (p, seed) or False / None: with a tuple (or False) the result is (dx, dropout(dx, p, seed) or None) -- the backward of the
        dropout that follows on the way down, written by the same kernel"""
        if drop is None:
            return K.layernorm_bwd(tape.s["x"], dy, tape.s["mr"], self.weight, _grad_buf(self.weight), _grad_buf(self.bias), dres)
        return K.layernorm_bwd(tape.s["x"], dy, tape.s["mr"], self.weight, _grad_buf(self.weight), _grad_buf(self.bias), dres,
                               drop=drop if drop else (0.0, 0))


class CausalSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.key = Linear(config.n_embd, config.n_embd)
        self.query = Linear(config.n_embd, config.n_embd)
        self.value = Linear(config.n_embd, config.n_embd)
        self.attn_drop = nn.Dropout(config.attn_pdrop)
        self.resid_drop = nn.Dropout(config.resid_pdrop)
        self.proj = Linear(config.n_embd, config.n_embd)
        mask = torch.tril(torch.ones(config.block_size, config.block_size))
        if hasattr(config, "n_unmasked"):
            mask[:config.n_unmasked, :config.n_unmasked] = 1
        self.register_buffer("mask", mask.view(1, 1, config.block_size, config.block_size))     # state_dict parity only
        self.n_head = config.n_head
        self._n_unmasked = int(getattr(config, "n_unmasked", 0) or 0)

    def _qkv_pack(self):
        """ONE bf16 operand for the three projections of the same input (stackgpt.py:46-48 of the reference runs them as three Linear
        layers): rows [Wk; Wq; Wv] for the forward GEMM (N = 3 C), its transpose [Wk^T | Wq^T | Wv^T] for the single input-gradient GEMM
        (K = 3 C) and the concatenated bias.  The buffers belong to the Linear pack registry's entries of the three layers, i.e. they are
        refreshed by the same launch that repacks every Linear after an optimizer step; the fp32 masters, their gradients and the
        state_dict stay three separate parameters"""
        from .layers import LINEAR_PACKS
        dev = self.key.weight.device
        ent = getattr(self, "_qkv", None)
        lp = getattr(self.key, "_lpack", None)
        # (a copied / moved module: the layers' pack entries must still BE slices of this module's buffers)
        if ent is None or ent["w"].device != dev or lp is None or lp["w"].data_ptr() != ent["w"].data_ptr():
            c = self.key.in_features
            wcat = torch.zeros(3 * c, c, dtype=torch.bfloat16, device=dev)
            wtcat = torch.zeros(c, 3 * c, dtype=torch.bfloat16, device=dev)
            bcat = torch.zeros(3 * c, dtype=torch.float32, device=dev)
            for i, lin in enumerate((self.key, self.query, self.value)):
                LINEAR_PACKS.register(lin, w=wcat[i * c:(i + 1) * c], wt=wtcat.view(-1)[i * c:], wt_ld=3 * c, bias_dst=bcat[i * c:(i + 1) * c])
            ent = self._qkv = {"w": wcat, "wt": wtcat, "b": bcat}
        for lin in (self.key, self.query, self.value):
            lin._w(torch.bfloat16)                      # (re)packs when the parameters changed: one launch for the whole optimizer group
        return ent

    def _qkv_fusable(self, x2d, hs):
        return (hs == 128 and x2d.dtype == torch.bfloat16 and os.environ.get("DVQ_QKV_FUSED", "1") != "0" and
                os.environ.get("DVQ_ATTN_V2", "1") != "0" and os.environ.get("DVQ_LINEAR_MULTIPACK", "1") != "0" and
                all(l.bias is not None and l.out_p == l.out_features for l in (self.key, self.query, self.value)))

    def fwd(self, x2d, b, t, tape, resid=None):
        c = x2d.shape[1]
        nh, hs = self.n_head, c // self.n_head
        if K.attn_causal_ok(x2d, nh, b, t) and not self._n_unmasked and self._qkv_fusable(x2d, hs):
            # key / query / value as ONE GEMM (N = 3 C) into [M, 3 C]; the attention kernels read the three column blocks with the row
            # pitch 3 C; the backward writes dk / dq / dv into the same layout and takes the input gradient as ONE GEMM over K = 3 C
            # (three GEMMs + two adds before).  DVQ_QKV_FUSED=0: three Linear layers
            m = x2d.shape[0]
            pk = self._qkv_pack()
            qkv = K.gemm_nt(x2d, pk["w"], m, 3 * c, c, c, c, 3 * c, bias=pk["b"], bias_mode=1).view(m, 3 * c)
            cols = (c, 0, 2 * c)                                 # first columns of q, k, v in [k | q | v]
            p_drop = self.attn_drop.p if self.training else 0.0
            seed = _next_seed() if p_drop > 0.0 else 0
            dm = None
            if tape is not None and p_drop > 0.0 and os.environ.get("DVQ_ATTN_DROP_MASK", "1") == "1":
                dm = K.attn_causal_drop_mask(x2d, b, t, nh)
            y, lse = K.attn_causal_fwd_fused(qkv, cols, b, t, nh, 1.0 / math.sqrt(hs), p_drop, seed, drop_mask=dm)
            out = self.proj.fwd(y, _child(tape, "proj"))
            if tape is not None:
                tape.s.update(qkv=qkv, cols=cols, fused=(p_drop, seed), y=y, lse=lse, drop_mask=dm, x=x2d, b=b, t=t)
            if resid is not None:
                return _drop_add(resid, out, self.resid_drop.p, self.training, tape, "rdrop")
            return _drop(out, self.resid_drop.p, self.training, tape, "rdrop")
        # (the key projection on the side stream beside the other two -- three independent GEMMs with partly empty last rounds --
        #  measured no gain: 82.0 vs 82.0 ms per step)
        k = self.key.fwd(x2d, _child(tape, "k"))
        q = self.query.fwd(x2d, _child(tape, "q"))
        v = self.value.fwd(x2d, _child(tape, "v"))
        fused = K.attn_causal_ok(x2d, nh, b, t) and not self._n_unmasked
        if fused:
            # one kernel per direction, scores stay in registers (csrc/attention.hip)
            p_drop = self.attn_drop.p if self.training else 0.0
            seed = _next_seed() if p_drop > 0.0 else 0
            # the forward leaves its keep decisions (1 bit per score, 14 MB per layer at the p6c18 geometry) for the three backward
            # kernels instead of each hashing every element again (19 of ~30 vector-ALU issue slots per score).  With the first-generation
            # kernels this bought nothing (84.14 vs 84.06 ms per step: they waited on LDS refills and barriers,
            # profiles/r04_attn_bwd_probe.txt); with the round-6 kernels (csrc/attention2.hip) the backward is 0.301 instead of 0.437 ms
            # per layer.  DVQ_ATTN_DROP_MASK=0: rehash
            dm = None
            if tape is not None and p_drop > 0.0 and os.environ.get("DVQ_ATTN_DROP_MASK", "1") == "1":
                dm = K.attn_causal_drop_mask(q, b, t, nh)
            y, lse = K.attn_causal_fwd(q, k, v, b, t, nh, 1.0 / math.sqrt(hs), p_drop, seed, drop_mask=dm)
            if tape is not None:
                tape.s.update(fused=(p_drop, seed), y=y, lse=lse, drop_mask=dm)
            p = pd = None
        else:
            s = torch.empty(b * nh * t * t, dtype=x2d.dtype, device=x2d.device)
            qf, kf = q.reshape(-1), k.reshape(-1)
            for h in range(nh):
                K.gemm_nt(qf[h * hs:], kf[h * hs:], t, t, hs, c, c, t, batch=b, sa=t * c, sb=t * c, sc=nh * t * t, out=s[h * t * t:])
            p = K.softmax_causal_(s, b * nh * t, t, t, 0, 1.0 / math.sqrt(hs))
            pd = _drop(p, self.attn_drop.p, self.training, tape, "adrop")
            vt = K.transpose(v, b, t, c).reshape(-1)                                  # [B, C, T]
            y = torch.empty(b * t, c, dtype=x2d.dtype, device=x2d.device)
            yf = y.reshape(-1)
            for h in range(nh):
                K.gemm_nt(pd[h * t * t:], vt[h * hs * t:], t, hs, t, t, t, c, batch=b, sa=nh * t * t, sb=c * t, sc=t * c, out=yf[h * hs:])
        out = self.proj.fwd(y, _child(tape, "proj"))
        if tape is not None:
            tape.s.update(q=q, k=k, v=v, p=p, pd=pd, b=b, t=t)
        if resid is not None:            # the block's residual stream: x + resid_drop(proj(..)) in one pass
            return _drop_add(resid, out, self.resid_drop.p, self.training, tape, "rdrop")
        return _drop(out, self.resid_drop.p, self.training, tape, "rdrop")

    def bwd(self, dout, tape, dout_dropped=None):
        """dout_dropped: resid_drop's backward already applied to dout (written by the LayerNorm backward that produced dout)"""
        s_ = tape.s
        if "qkv" in s_:
            return self._bwd_qkv(dout, tape, dout_dropped)
        q, k, v, p, pd, b, t = s_["q"], s_["k"], s_["v"], s_["p"], s_["pd"], s_["b"], s_["t"]
        c = q.shape[1]
        nh, hs = self.n_head, c // self.n_head
        dout = dout_dropped if dout_dropped is not None else _drop_bwd(dout, tape, "rdrop")
        dy = self.proj.bwd(dout, tape.child("proj"))
        if "fused" in s_:
            p_drop, seed = s_["fused"]
            dq, dk, dv = K.attn_causal_bwd(q, k, v, s_["y"], dy, s_["lse"], b, t, nh, 1.0 / math.sqrt(hs), p_drop, seed,
                                           drop_mask=s_.get("drop_mask"))
            dx = self.query.bwd(dq, tape.child("q"))
            if os.environ.get("DVQ_LINEAR_ADDEND", "0") == "1":
                # accumulating in the GEMM epilogues (dvq_gemm_nt_res) instead of two add kernels measured SLOWER: 82.2 vs 80.9 ms per
                # step -- the HBM-bound adds overlap the weight-gradient GEMMs on the side stream, the epilogue work does not
                dx = self.key.bwd(dk, tape.child("k"), addend=dx)
                return self.value.bwd(dv, tape.child("v"), addend=dx)
            dx = K.add(dx, self.key.bwd(dk, tape.child("k")))
            return K.add(dx, self.value.bwd(dv, tape.child("v")))
        dyf, vf, qf = dy.reshape(-1), v.reshape(-1), q.reshape(-1)
        dp = torch.empty_like(p)
        dv32 = torch.zeros(b * t * c, dtype=torch.float32, device=dy.device)
        for h in range(nh):
            K.gemm_nt(dyf[h * hs:], vf[h * hs:], t, t, hs, c, c, t, batch=b, sa=t * c, sb=t * c, sc=nh * t * t, out=dp[h * t * t:])
            K.gemm_tn(pd[h * t * t:], dyf[h * hs:], t, t, hs, t, c, c, batch=b, sa=nh * t * t, sb=t * c, sc=t * c, out=dv32[h * hs:])
        dp = _drop_bwd(dp, tape, "adrop")
        ds = K.softmax_rows_bwd(p, dp, b * nh * t, t, 1.0 / math.sqrt(hs))
        kt = K.transpose(k, b, t, c).reshape(-1)
        dq = torch.empty(b * t, c, dtype=dy.dtype, device=dy.device)
        dqf = dq.reshape(-1)
        dk32 = torch.zeros(b * t * c, dtype=torch.float32, device=dy.device)
        for h in range(nh):
            K.gemm_nt(ds[h * t * t:], kt[h * hs * t:], t, hs, t, t, t, c, batch=b, sa=nh * t * t, sb=c * t, sc=t * c, out=dqf[h * hs:])
            K.gemm_tn(ds[h * t * t:], qf[h * hs:], t, t, hs, t, c, c, batch=b, sa=nh * t * t, sb=t * c, sc=t * c, out=dk32[h * hs:])
        dx = self.query.bwd(dq, tape.child("q"))
        dx = K.add(dx, self.key.bwd(K.cast(dk32.view(b * t, c), dy.dtype), tape.child("k")))
        dx = K.add(dx, self.value.bwd(K.cast(dv32.view(b * t, c), dy.dtype), tape.child("v")))
        return dx


    def _bwd_qkv(self, dout, tape, dout_dropped):
        """backward of the fused-projection forward: attention backward into [dk | dq | dv] ([M, 3 C]), ONE input-gradient GEMM over
        K = 3 C, the three weight / bias gradients from column blocks of that matrix (side stream, like Linear.bwd)"""
        s_ = tape.s
        qkv, cols, x2d, b, t = s_["qkv"], s_["cols"], s_["x"], s_["b"], s_["t"]
        m, c3 = qkv.shape
        c = c3 // 3
        nh, hs = self.n_head, c // self.n_head
        dout = dout_dropped if dout_dropped is not None else _drop_bwd(dout, tape, "rdrop")
        dy = self.proj.bwd(dout, tape.child("proj"))
        p_drop, seed = s_["fused"]
        dqkv = K.attn_causal_bwd_fused(qkv, cols, s_["y"], dy, s_["lse"], b, t, nh, 1.0 / math.sqrt(hs), p_drop, seed,
                                       drop_mask=s_.get("drop_mask"))
        pk = self._qkv_pack()
        dflat = dqkv.view(-1)

        def wgrad():
            for i, lin in enumerate((self.key, self.query, self.value)):
                K.gemm_tn(dflat[i * c:], x2d, m, c, c, c3, c, c, out=_grad_buf(lin.weight), colsum=_grad_buf(lin.bias))

        dx = K.gemm_nt(dqkv, pk["wt"], m, c, c3, c3, c3, c)
        if m >= 1024 and rt.side_wgrad_enabled() and os.environ.get("DVQ_LINEAR_SIDE", "1") != "0":
            rt.run_on_side(wgrad, dqkv, x2d)
        else:
            wgrad()
        return dx.view(m, c)


class Block(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.ln1 = LayerNorm(config.n_embd)
        self.ln2 = LayerNorm(config.n_embd)
        self.attn = CausalSelfAttention(config)
        self.mlp = nn.Sequential(Linear(config.n_embd, 4 * config.n_embd), nn.GELU(), Linear(4 * config.n_embd, config.n_embd),
                                 nn.Dropout(config.resid_pdrop))

    def fwd(self, x, b, t, tape):
        # (residual adds and resid_drop ride in one kernel each; the backward's residual adds ride in the LayerNorm backward)
        x1 = self.attn.fwd(self.ln1.fwd(x, _child(tape, "ln1")), b, t, _child(tape, "attn"), resid=x)
        hid = self.mlp[0].fwd(self.ln2.fwd(x1, _child(tape, "ln2")), _child(tape, "fc1"))
        m = self.mlp[2].fwd(K.gelu(hid), _child(tape, "fc2"))
        if tape is not None:
            tape.s["hid"] = hid
        return _drop_add(x1, m, self.mlp[3].p, self.training, tape, "mdrop")

    def bwd(self, d, tape, d_dropped=None, next_drop=None):
        """d: gradient of the block's output.  The two dropouts on the way down (the MLP's, the attention projection's) need
        dropout(gradient) of tensors a LayerNorm backward has just written: that kernel emits the dropped copy too (fused: 2 x 24
        stand-alone dropout passes of a p6c18 train step, 2.6 ms, disappear).  d_dropped: the MLP dropout's backward of d, already
        made by the block above; next_drop: (p, seed) of the MLP dropout of the block BELOW, or False -- then the result is the pair
        (dx, dropout(dx) or None) for that block.  DVQ_FUSE_DROP_BWD=0: stand-alone passes"""
        fuse = _fuse_drop_bwd()
        dm = d_dropped if d_dropped is not None else _drop_bwd(d, tape, "mdrop")
        dact = self.mlp[2].bwd(dm, tape.child("fc2"))
        dh2 = self.mlp[0].bwd(K.gelu_bwd(tape.s["hid"], dact), tape.child("fc1"))
        rd = tape.child("attn").s.get("rdrop") if fuse else None
        if rd is not None:
            dx1, dx1d = self.ln2.bwd(dh2, tape.child("ln2"), dres=d, drop=rd)
        else:
            dx1, dx1d = self.ln2.bwd(dh2, tape.child("ln2"), dres=d), None
        dh1 = self.attn.bwd(dx1, tape.child("attn"), dout_dropped=dx1d)
        if next_drop is None:
            return self.ln1.bwd(dh1, tape.child("ln1"), dres=dx1)
        return self.ln1.bwd(dh1, tape.child("ln1"), dres=dx1, drop=next_drop if fuse else False)


def _attn_append(attn, x2d, b, n, cache, t0):
    """CausalSelfAttention over a K/V cache: append n rows per sequence (rows t0 .. t0+n-1) and attend causally over rows
    [0, t0+n).  cache = (k [B,Tmax,C], v [B,Tmax,C]).  n == 1 uses the single-query kernel, otherwise per-head GEMMs."""
    c = x2d.shape[1]
    nh, hs = attn.n_head, c // attn.n_head
    k = attn.key.fwd(x2d, None)
    q = attn.query.fwd(x2d, None)
    v = attn.value.fwd(x2d, None)
    kc, vc = cache
    kc[:, t0:t0 + n].copy_(k.view(b, n, c))
    vc[:, t0:t0 + n].copy_(v.view(b, n, c))
    t = t0 + n
    scale = 1.0 / math.sqrt(hs)
    if n == 1:
        y = K.attn_decode(q, kc, vc, nh, t, scale)
    else:
        tmax = kc.shape[1]
        # attend over tk = t rounded up to 8 cache rows: rows [t, tk) exist in the cache (zeros or finite values of an earlier run), lie in
        # every query's causal future and get probability exactly 0 -- and both products stay on the MFMA kernels (an odd t sent the
        # P V product to the naive kernel: 180 us per call, 2.8 % of a sampling run)
        tk = min(tmax, (t + 7) // 8 * 8)
        if tk % 8 != 0:
            tk = t
        if tk > t:
            # probability 0 times a NaN / Inf left in rows [t, tk) by an overflowed earlier run (or a cache that was not
            # zero-initialised) would still poison every query through the P V product: clear the few padding rows (prefill only)
            kc[:, t:tk].zero_()
            vc[:, t:tk].zero_()
        s = torch.empty(b * nh * n * tk, dtype=x2d.dtype, device=x2d.device)
        qf, kf = q.reshape(-1), kc.reshape(-1)
        for h in range(nh):
            K.gemm_nt(qf[h * hs:], kf[h * hs:], n, tk, hs, c, c, tk, batch=b, sa=n * c, sb=tmax * c, sc=nh * n * tk, out=s[h * n * tk:])
        K.softmax_causal_(s, b * nh * n, tk, n, t0, scale)
        # P [n, tk] x V [tk, hs]: V^T per head from the cache rows [0, tk)
        vt = K.transpose(vc[:, :tk].contiguous(), b, tk, c).reshape(-1)           # [B, C, tk]
        y = torch.empty(b * n, c, dtype=x2d.dtype, device=x2d.device)
        yf = y.reshape(-1)
        tp8 = tk % 8 == 0
        for h in range(nh):
            K.gemm_nt(s[h * n * tk:], vt[h * hs * tk:], n, hs, tk, tk, tk, c, batch=b, sa=nh * n * tk, sb=c * tk, sc=n * c, out=yf[h * hs:],
                      impl=0 if tp8 else 1)
    return attn.proj.fwd(y, None)


def _block_append(blk, x2d, b, n, cache, t0):
    a = _attn_append(blk.attn, blk.ln1.fwd(x2d, None), b, n, cache, t0)
    x1 = K.add(x2d, a)
    m = blk.mlp[2].fwd(K.gelu(blk.mlp[0].fwd(blk.ln2.fwd(x1, None), None)), None)
    return K.add(x1, m)


def _block_append_dev(blk, x2d, cache, t_dev):
    """_block_append for ONE new row per sequence whose cache row index lives in device memory (t_dev, int64 [1]): nothing in the
    launch sequence depends on the position, so the whole pass can be captured once as a hipGraph and replayed per token"""
    attn = blk.attn
    c = x2d.shape[1]
    h = blk.ln1.fwd(x2d, None)
    k = attn.key.fwd(h, None)
    q = attn.query.fwd(h, None)
    v = attn.value.fwd(h, None)
    y = K.attn_decode_dev(q, k, v, cache[0], cache[1], attn.n_head, t_dev, 1.0 / math.sqrt(c // attn.n_head))
    x1 = K.add(x2d, attn.proj.fwd(y, None))
    m = blk.mlp[2].fwd(K.gelu(blk.mlp[0].fwd(blk.ln2.fwd(x1, None), None)), None)
    return K.add(x1, m)


class DecodeState:
    """K/V caches of both transformers + the position-transformer hidden rows of one sampling run (eval mode, no dropout).
    The position transformer's row r is (content_r, position_r [, segment_r]); the content transformer's row r is that
    hidden row plus the embedding of an `update` position token (stackgpt.py:189-196, 252-339)."""

    def __init__(self, gpt, batch, max_rows):
        cd = rt.compute_dtype()
        dev, c = gpt.pos_emb.device, gpt.config.n_embd
        self.gpt, self.b, self.max_rows = gpt, batch, max_rows
        mk = lambda: (torch.zeros(batch, max_rows, c, dtype=cd, device=dev), torch.zeros(batch, max_rows, c, dtype=cd, device=dev))
        self.pos_cache = [mk() for _ in gpt.position_transformer]
        self.con_cache = [mk() for _ in gpt.content_transformer]
        self.hidden = torch.zeros(batch, max_rows, c, dtype=cd, device=dev)
        self.rows_pos = 0
        self.rows_con = 0
        # single-row steps: row counters in device memory + one captured hipGraph per (transformer, position table) -- the
        # sampler is launch-bound (24 layers x ~12 small kernels per token), a replay costs one launch.  DVQ_DECODE_GRAPH=0: eager
        self.t_pos = torch.zeros(1, dtype=torch.long, device=dev)
        self.t_con = torch.zeros(1, dtype=torch.long, device=dev)
        self.use_graph = os.environ.get("DVQ_DECODE_GRAPH", "1") != "0"
        self._steps = {}
        self._stacks = {}
        self._sig = self._weights_signature()

    def _weights_signature(self):
        g = self.gpt
        ps = [g.position_transformer[0].attn.key.weight, g.content_transformer[-1].mlp[2].weight, g.content_head[1].weight]
        return tuple((rt.param_epoch(p_), p_._version, p_.data_ptr()) for p_ in ps)

    def reset(self):
        """start a new sampling run on the same buffers (rows are always written before they are read)"""
        self.rows_pos = self.rows_con = 0
        self.t_pos.zero_()
        self.t_con.zero_()
        sig = self._weights_signature()
        if sig != self._sig:              # the captured graphs point at the previous compute-dtype weight copies
            self._steps.clear()
            self._stacks.clear()
            self._sig = sig

    def reset_content(self):
        self.rows_con = 0
        self.t_con.zero_()

    def check(self):
        """raise if a device-wide barrier of the persistent token-step kernel timed out since the last check (its workgroups were
        not all resident, e.g. on a shared GPU): the rows written since are invalid.  One 16-byte read-back per transformer --
        called once per sampling run, not per token.  After the error the fused kernel is switched off for this state (the next
        runs use the per-kernel token steps)."""
        for which, ent in list(self._stacks.items()):
            try:
                K.decode_stack_status(ent["scratch"], self.b, self.gpt.config.n_embd, ent["f"])
            except RuntimeError:
                self._stacks.clear()
                self._steps.clear()
                self._no_stack = True
                raise

    # ---- all blocks of a transformer in one persistent kernel (dvq_decode_stack) ---------------------------------------------------
    def _stack(self, which):
        """(device table of the blocks' weight / cache pointers, scratch, tensors kept alive) of the position ('pos') or content
        ('con') transformer, or None when the fused kernel does not apply (fp32 runs, > 64 sequences, DVQ_DECODE_STACK=0)"""
        g = self.gpt
        blocks, caches = (g.position_transformer, self.pos_cache) if which == "pos" else (g.content_transformer, self.con_cache)
        c, cd = g.config.n_embd, rt.compute_dtype()
        nh = blocks[0].attn.n_head
        if (getattr(self, "_no_stack", False) or os.environ.get("DVQ_DECODE_STACK", "1") == "0" or cd != torch.bfloat16 or self.b > 64 or c % 32 or c > 2048 or (c // nh) % 8 or
                c // nh > 256 or self.max_rows > 12000):
            return None
        ent = self._stacks.get(which)
        if ent is not None and ent["sig"] == self._sig:
            return ent
        import ctypes
        from ._lib import DecodeLayer
        arr = (DecodeLayer * len(blocks))()
        keep = []
        for i, (blk, cache) in enumerate(zip(blocks, caches)):
            lins = [blk.attn.query, blk.attn.key, blk.attn.value, blk.attn.proj, blk.mlp[0], blk.mlp[2]]
            if any(l.out_p != l.out_features for l in lins):
                return None
            wb = [l._w(cd) for l in lins]
            keep.append(wb)
            ptr = lambda t_: 0 if t_ is None else t_.data_ptr()
            arr[i] = DecodeLayer(*[ptr(w) for w, _ in wb], *[ptr(b_) for _, b_ in wb], ptr(blk.ln1.weight), ptr(blk.ln1.bias),
                                 ptr(blk.ln2.weight), ptr(blk.ln2.bias), cache[0].data_ptr(), cache[1].data_ptr())
        dev = self.hidden.device
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        f = blocks[0].mlp[0].out_features
        ent = {"sig": self._sig, "table": table, "n": len(blocks), "nh": nh, "f": f, "eps": blocks[0].ln1.eps,
               "scratch": K.decode_stack_scratch(self.b, c, f, dev), "keep": keep, "host": arr}
        self._stacks[which] = ent
        return ent

    def _run_blocks(self, which, x, t_dev):
        ent = self._stack(which)
        if ent is None:
            blocks, caches = (self.gpt.position_transformer, self.pos_cache) if which == "pos" else (self.gpt.content_transformer, self.con_cache)
            for blk, cache in zip(blocks, caches):
                x = _block_append_dev(blk, x, cache, t_dev)
            return x
        return K.decode_stack(ent["table"], ent["n"], x.contiguous(), ent["nh"], ent["f"], self.max_rows, t_dev, ent["eps"], ent["scratch"],
                              int(os.environ.get("DVQ_DECODE_WGS", "0")), table_host=ent["host"])

    def _step(self, key, body, inputs):
        """run `body(*static_inputs)`: eagerly the first time (creates weight caches, kernel attributes), then captured once and
        replayed"""
        ent = self._steps.get(key)
        if ent is None:
            ent = self._steps[key] = {"static": [None if t is None else t.clone() for t in inputs], "graph": None, "calls": 0}
        for st_, t in zip(ent["static"], inputs):
            if st_ is not None:
                st_.copy_(t)
        ent["calls"] += 1
        if not self.use_graph or ent["calls"] == 1:
            return body(*ent["static"])
        if ent["graph"] is None:
            # replayed as a launch list (csrc/cmdlist.hip: the recorded launches re-issued on this lane's stream, ~2 us of host time
            # each) unless DVQ_DECODE_REPLAY=graph: hipGraphLaunch of a ~130-node token step costs ~0.8 ms of host time on ROCm 7.2,
            # which capped concurrent lanes at 1250 token steps / s whatever their number (profiles/r06_sampler_lanes.txt)
            as_list = os.environ.get("DVQ_DECODE_REPLAY", "list") == "list"
            gr = torch.cuda.CUDAGraph(keep_graph=True) if as_list else torch.cuda.CUDAGraph()
            # thread-local capture mode: another sampling lane (Dualformer.sample_many runs one host thread per stream) may launch and
            # allocate on ITS stream while this one records
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                ent["out"] = body(*ent["static"])
            ent["graph"], ent["list"] = gr, None
            if as_list:
                from ._lib import DvqError
                try:
                    ent["list"] = K.CmdList(gr)
                except DvqError:          # a node kind the list cannot re-issue (a torch memcpy): this body replays as a hipGraph
                    ent["list"] = None
        if ent.get("list") is not None:
            cur = torch.cuda.current_stream()
            ent["list"].replay(cur, cur)
        else:
            ent["graph"].replay()
        return ent["out"].clone()

    def _position_row_body(self, pos_table):
        g, b = self.gpt, self.b

        def body(content_tok, pos_tok, seg_tok):
            pieces = [(g.content_emb.weight, content_tok, 0, None, False), (g.pos_emb, self.t_pos, 0, None, True),
                      (pos_table, pos_tok, 0, None, False)]
            if seg_tok is not None:
                pieces.append((g.seg_emb.weight, seg_tok, 0, None, False))
            x = g._embed(pieces, b, 1, None, "").view(b, -1)
            x = self._run_blocks("pos", x, self.t_pos)
            K.rows_dev(x, self.hidden, self.t_pos, True)
            logits = g._head(g.position_head, x, None, "ph")[:, : g.config.fine_position_size].float()
            self.t_pos.add_(1)
            return logits
        return body

    def _content_row_body(self, upd_table):
        g, b = self.gpt, self.b

        def body(upd_tok):
            upd = g._embed([(upd_table, upd_tok, 0, None, False)], b, 1, None, "").view(b, -1)
            x = K.add(K.rows_dev(torch.empty_like(upd), self.hidden, self.t_con, False), upd)
            x = self._run_blocks("con", x, self.t_con)
            logits = g._head(g.content_head, x, None, "ch")[:, : g.config.vocab_size].float()
            self.t_con.add_(1)
            return logits
        return body

    @torch.no_grad()
    def position_rows(self, content_tok, pos_tok, pos_table, pos_pad, seg_tok=None):
        """append n rows (tokens [B,n]) to the position transformer; -> position logits of the last appended row [B,V]"""
        g, b = self.gpt, self.b
        n, t0 = content_tok.shape[1], self.rows_pos
        assert t0 + n <= self.max_rows, "DecodeState: increase max_rows"
        if n == 1:
            use_seg = g.activate_segment and seg_tok is not None
            out = self._step(("pos", id(pos_table), use_seg), self._position_row_body(pos_table),
                             [content_tok.contiguous(), pos_tok.contiguous(), seg_tok.contiguous() if use_seg else None])
            self.rows_pos = t0 + 1
            return out
        ar = torch.arange(t0, t0 + n, device=content_tok.device)
        pieces = [(g.content_emb.weight, content_tok.contiguous(), 0, None, False), (g.pos_emb, ar, 0, None, True),
                  (pos_table, pos_tok.contiguous(), 0, None, False)]
        if g.activate_segment and seg_tok is not None:
            pieces.append((g.seg_emb.weight, seg_tok.contiguous(), 0, None, False))
        x = g._embed(pieces, b, n, None, "").view(b * n, -1)
        for blk, cache in zip(g.position_transformer, self.pos_cache):
            x = _block_append(blk, x, b, n, cache, t0)
        self.hidden[:, t0:t0 + n].copy_(x.view(b, n, -1))
        self.rows_pos = t0 + n
        self.t_pos.fill_(self.rows_pos)
        last = x.view(b, n, -1)[:, -1].contiguous()
        return g._head(g.position_head, last, None, "ph")[:, : g.config.fine_position_size].float()

    @torch.no_grad()
    def content_rows(self, upd_tok, upd_table):
        """append the next n = upd_tok.shape[1] rows to the content transformer: hidden rows + embedding of the update position
        tokens; -> content logits of the last appended row [B,V]"""
        g, b = self.gpt, self.b
        n, t0 = upd_tok.shape[1], self.rows_con
        assert t0 + n <= self.rows_pos, "content rows cannot run ahead of the position transformer"
        if n == 1:
            out = self._step(("con", id(upd_table)), self._content_row_body(upd_table), [upd_tok.contiguous()])
            self.rows_con = t0 + 1
            return out
        upd = g._embed([(upd_table, upd_tok.contiguous(), 0, None, False)], b, n, None, "")
        x = K.add(self.hidden[:, t0:t0 + n].contiguous().view(b * n, -1), upd.view(b * n, -1))
        for blk, cache in zip(g.content_transformer, self.con_cache):
            x = _block_append(blk, x, b, n, cache, t0)
        self.rows_con = t0 + n
        self.t_con.fill_(self.rows_con)
        last = x.view(b, n, -1)[:, -1].contiguous()
        return g._head(g.content_head, last, None, "ch")[:, : g.config.vocab_size].float()


class _Embedding(nn.Embedding):
    """nn.Embedding parameters (incl. padding_idx bookkeeping for state_dict / init parity); lookups run on dvq_embed_*"""


class StackGPT(nn.Module):
    def __init__(self, vocab_size, coarse_position_size, fine_position_size, segment_size=-1, block_size=None, position_layer=12,
                 content_layer=12, n_head=8, n_embd=256, embd_pdrop=0., resid_pdrop=0., attn_pdrop=0., content_pad_code=1025,
                 coarse_position_pad_code=257, fine_position_pad_code=1025, activate_pad_ignore=True):
        super().__init__()
        config = StackGPTConfig(vocab_size=vocab_size, coarse_position_size=coarse_position_size, fine_position_size=fine_position_size,
                                block_size=block_size, embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop,
                                position_layer=position_layer, content_layer=content_layer, n_head=n_head, n_embd=n_embd, n_unmasked=0)
        self.activate_segment = segment_size > 0
        self.activate_pad_ignore = activate_pad_ignore
        self.coarse_position_pad_code, self.fine_position_pad_code = coarse_position_pad_code, fine_position_pad_code
        self.content_pad_code = content_pad_code
        self.block_size = config.block_size
        self.config = config
        self.content_coarse_pos_emb = _Embedding(coarse_position_size, n_embd, padding_idx=coarse_position_pad_code)
        self.content_fine_pos_emb = _Embedding(fine_position_size, n_embd, padding_idx=fine_position_pad_code)
        self.content_emb = _Embedding(vocab_size, n_embd, padding_idx=content_pad_code)
        self.pos_emb = nn.Parameter(torch.zeros(1, config.block_size, n_embd))
        if self.activate_segment:
            self.seg_emb = _Embedding(segment_size, n_embd)
        self.drop = nn.Dropout(embd_pdrop)
        self.position_transformer = nn.Sequential(*[Block(config) for _ in range(position_layer)])
        self.content_transformer = nn.Sequential(*[Block(config) for _ in range(content_layer)])
        self.position_head = nn.Sequential(LayerNorm(n_embd), Linear(n_embd, fine_position_size, bias=False))
        self.content_head = nn.Sequential(LayerNorm(n_embd), Linear(n_embd, vocab_size, bias=False))
        self.apply(self._init_weights)
        self._grad_hook = None             # set by the Trainer under data parallelism (per-block gradient exchange, see _run_bwd)

    def get_block_size(self):
        return self.block_size

    def _init_weights(self, module):
        if isinstance(module, (Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    # ---- shared pieces -------------------------------------------------------------------------------------------
    def _pad_t(self, t):
        return -(-t // 8) * 8

    def _embed(self, pieces, b, t_pad, tape, key):
        """pieces: list of (table parameter, idx [B,len] or [len], t0, padding_idx, shared_over_batch); -> [B,t_pad,C] sum"""
        cd = rt.compute_dtype()
        c = self.config.n_embd
        out = torch.zeros(b, t_pad, c, dtype=cd, device=self.pos_emb.device)
        for table, idx, t0, pad, shared in pieces:
            K.embed_gather(idx, table.detach().reshape(-1, c), out, t0, True, bstride=0 if shared else None)
        if tape is not None:
            tape.s[key] = pieces
        return out

    def _embed_bwd(self, g3d, tape, key):
        c = self.config.n_embd
        for table, idx, t0, pad, shared in tape.s[key]:
            K.embed_scatter_add(idx, g3d, _grad_buf(table).view(-1, c), t0, padding_idx=-1 if pad is None else pad,
                                bstride=0 if shared else None)

    def _run(self, blocks, x2d, b, t, tape, name):
        for i, blk in enumerate(blocks):
            x2d = blk.fwd(x2d, b, t, _child(tape, f"{name}{i}"))
        return x2d

    def _run_bwd(self, blocks, g, tape, name):
        hook = getattr(self, "_grad_hook", None)
        gd = None
        for i in reversed(range(len(blocks))):
            # (p, seed) of the MLP dropout of the block below: its backward is written by this block's last LayerNorm backward
            nd = (tape.child(f"{name}{i - 1}").s.get("mdrop") or False) if i > 0 else False
            g, gd = blocks[i].bwd(g, tape.child(f"{name}{i}"), d_dropped=gd, next_drop=nd)
            if hook is not None:           # data parallel: this block's gradients are final -> their all-reduce starts now
                hook(list(blocks[i].parameters()))
        return g

    def _head(self, head, x2d, tape, name):
        return head[1].fwd(head[0].fwd(x2d, _child(tape, name + "ln")), _child(tape, name + "fc"))

    def _head_bwd(self, head, dlogits, tape, name):
        return head[0].bwd(head[1].bwd(dlogits, tape.child(name + "fc")), tape.child(name + "ln"))

    # ---- training / teacher-forced forward (stackgpt.py:175-232) ----------------------------------------------------
    def fwd(self, coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg, tape):
        """-> (position_logits2d [B*Tp, Vp_pad], content_logits2d [B*Tp, Vc_pad], B, T, T_pad)"""
        b, lc = coarse_position.shape
        lf = fine_position.shape[1]
        t = lc + lf - 1
        tp = self._pad_t(t)
        dev = coarse_content.device
        content = torch.cat([coarse_content, fine_content], dim=1)[:, :-1].contiguous()
        ar = torch.arange(tp, device=dev)
        pieces = [(self.content_emb.weight, content, 0, self.content_pad_code, False),
                  (self.content_coarse_pos_emb.weight, coarse_position.contiguous(), 0, self.coarse_position_pad_code, False),
                  (self.pos_emb, ar, 0, None, True)]
        if lf > 1:
            pieces.append((self.content_fine_pos_emb.weight, fine_position[:, :-1].contiguous(), lc, self.fine_position_pad_code, False))
        if self.activate_segment:
            seg = torch.cat([coarse_seg, fine_seg], dim=1)[:, :-1].contiguous()
            pieces.append((self.seg_emb.weight, seg, 0, None, False))
        x = self._embed(pieces, b, tp, tape, "emb_in").view(b * tp, -1)
        x = _drop(x, self.drop.p, self.training, tape, "edrop")
        pos_hidden = self._run(self.position_transformer, x, b, tp, tape, "p")
        upd = [(self.content_fine_pos_emb.weight, fine_position.contiguous(), lc - 1, self.fine_position_pad_code, False)]
        if lc > 1:
            upd.append((self.content_coarse_pos_emb.weight, coarse_position[:, 1:].contiguous(), 0, self.coarse_position_pad_code, False))
        cin = K.add(pos_hidden, self._embed(upd, b, tp, tape, "emb_upd").view(b * tp, -1))
        con_hidden = self._run(self.content_transformer, cin, b, tp, tape, "c")
        content_logits = self._head(self.content_head, con_hidden, tape, "ch")
        position_logits = self._head(self.position_head, pos_hidden, tape, "ph")
        return position_logits, content_logits, b, t, tp

    def bwd(self, d_position_logits, d_content_logits, tape, b, tp):
        g_con = self._head_bwd(self.content_head, d_content_logits, tape, "ch")
        g_cin = self._run_bwd(self.content_transformer, g_con, tape, "c")
        self._embed_bwd(g_cin.view(b, tp, -1), tape, "emb_upd")
        g_pos = K.add(g_cin, self._head_bwd(self.position_head, d_position_logits, tape, "ph"))
        g_x = self._run_bwd(self.position_transformer, g_pos, tape, "p")
        g_x = _drop_bwd(g_x, tape, "edrop")
        self._embed_bwd(g_x.view(b, tp, -1), tape, "emb_in")

    def forward(self, coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg, content_target=None,
                coarse_position_target=None, fine_position_target=None, **ignorekwargs):
        params = [p for p in self.parameters() if p.requires_grad]
        with_loss = content_target is not None and coarse_position_target is not None and fine_position_target is not None
        if not with_loss:
            with torch.no_grad():
                pl, cl, b, t, tp = self.fwd(coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg, None)
            vp, vc = self.config.fine_position_size, self.config.vocab_size
            return {"position_logits": pl.view(b, tp, -1)[:, :t, :vp].float(), "content_logits": cl.view(b, tp, -1)[:, :t, :vc].float()}
        pos, con, cpl, fpl = _StackGPTLossFn.apply(self, torch.is_grad_enabled() and len(params) > 0,
                                                   (coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg,
                                                    content_target, coarse_position_target, fine_position_target), *params)
        return {"position_loss": pos, "content_loss": con, "coarse_position_loss": cpl, "fine_position_loss": fpl}

    # ---- prefix passes used by the sampler (stackgpt.py:234-339): forward only, no cache (the reference recomputes the
    # whole prefix at every step; so does this first version -- see DESIGN.md "next") ----------------------------------
    def _prefix_hidden(self, content, pos_pieces, seg, drop):
        """content [B,T]; pos_pieces: list of (table, idx [B,len], t0, pad); seg [B,T] or None -> position_hidden [B*Tp, C]"""
        b, t = content.shape
        tp = self._pad_t(t)
        ar = torch.arange(tp, device=content.device)
        pieces = [(self.content_emb.weight, content.contiguous(), 0, self.content_pad_code, False), (self.pos_emb, ar, 0, None, True)]
        pieces += [(tb, ix.contiguous(), t0, pad, False) for tb, ix, t0, pad in pos_pieces if ix.shape[1] > 0]
        if self.activate_segment and seg is not None:
            pieces.append((self.seg_emb.weight, seg.contiguous(), 0, None, False))
        x = self._embed(pieces, b, tp, None, "").view(b * tp, -1)
        if drop:
            x = _drop(x, self.drop.p, self.training, None, "")
        return self._run(self.position_transformer, x, b, tp, None, "p"), b, t, tp

    def _content_from_hidden(self, hidden2d, upd_pieces, b, t, tp):
        upd = self._embed([(tb, ix.contiguous(), t0, pad, False) for tb, ix, t0, pad in upd_pieces if ix.shape[1] > 0], b, tp, None, "")
        ch = self._run(self.content_transformer, K.add(hidden2d, upd.view(b * tp, -1)), b, tp, None, "c")
        logits = self._head(self.content_head, ch, None, "ch").view(b, tp, -1)[:, :t, : self.config.vocab_size].float()
        return ch.view(b, tp, -1)[:, :t], logits

    def _as2d(self, hidden, b, t, tp):
        """a hidden state handed back by the caller ([B,T,C]) -> padded [B*Tp, C]"""
        if tp == t:
            return hidden.reshape(b * t, -1).contiguous()
        out = torch.zeros(b, tp, hidden.shape[-1], dtype=hidden.dtype, device=hidden.device)
        out[:, :t] = hidden
        return out.view(b * tp, -1)

    @torch.no_grad()
    def sample_coarse_position(self, coarse_content, coarse_position, coarse_seg):
        cpe, pad = self.content_coarse_pos_emb.weight, self.coarse_position_pad_code
        h, b, t, tp = self._prefix_hidden(coarse_content, [(cpe, coarse_position, 0, pad)], coarse_seg, drop=False)
        logits = self._head(self.position_head, h, None, "ph").view(b, tp, -1)[:, :t, : self.config.fine_position_size].float()
        return h.view(b, tp, -1)[:, :t], logits

    @torch.no_grad()
    def sample_coarse_content(self, coarse_content=None, coarse_position=None, coarse_seg=None, position_hidden=None):
        cpe, pad = self.content_coarse_pos_emb.weight, self.coarse_position_pad_code
        if position_hidden is None:
            seg = coarse_seg[:, :-1] if coarse_seg is not None else None
            h, b, t, tp = self._prefix_hidden(coarse_content, [(cpe, coarse_position[:, :-1], 0, pad)], seg, drop=False)
        else:
            b, t = position_hidden.shape[0], position_hidden.shape[1]
            tp = self._pad_t(t)
            h = self._as2d(position_hidden, b, t, tp)
        return self._content_from_hidden(h, [(cpe, coarse_position[:, 1:], 0, pad)], b, t, tp)

    @torch.no_grad()
    def sample_fine_position(self, coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg):
        lc = coarse_position.shape[1]
        content = torch.cat([coarse_content, fine_content], dim=1)
        seg = None
        if self.activate_segment:
            seg = torch.cat([coarse_seg, fine_seg], dim=1) if fine_seg is not None else coarse_seg
        h, b, t, tp = self._prefix_hidden(content, [(self.content_coarse_pos_emb.weight, coarse_position, 0, self.coarse_position_pad_code),
                                                    (self.content_fine_pos_emb.weight, fine_position, lc, self.fine_position_pad_code)],
                                          seg, drop=True)
        logits = self._head(self.position_head, h, None, "ph").view(b, tp, -1)[:, :t, : self.config.fine_position_size].float()
        return h.view(b, tp, -1)[:, :t], logits

    @torch.no_grad()
    def sample_fine_content(self, coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg,
                            position_hidden=None):
        lc = coarse_position.shape[1]
        cpe, fpe = self.content_coarse_pos_emb.weight, self.content_fine_pos_emb.weight
        if position_hidden is None:
            content = torch.cat([coarse_content, fine_content], dim=1)
            seg = None
            if self.activate_segment:
                seg = torch.cat([coarse_seg, fine_seg], dim=1) if fine_seg is not None else coarse_seg
            h, b, t, tp = self._prefix_hidden(content, [(cpe, coarse_position, 0, self.coarse_position_pad_code),
                                                        (fpe, fine_position[:, :-1], lc, self.fine_position_pad_code)], seg, drop=True)
        else:
            b, t = position_hidden.shape[0], position_hidden.shape[1]
            tp = self._pad_t(t)
            h = self._as2d(position_hidden, b, t, tp)
        return self._content_from_hidden(h, [(cpe, coarse_position, 0, self.coarse_position_pad_code),
                                             (fpe, fine_position[:, 1:], lc, self.fine_position_pad_code)], b, t, tp)

    def _targets(self, b, t, tp, lc, content_target, coarse_position_target, fine_position_target, dev):
        """full-length [B*Tp] target rows per loss; rows a loss does not cover carry that loss's ignore index"""
        ign_c = self.content_pad_code if self.activate_pad_ignore else -100
        tc = torch.full((b, tp), ign_c, dtype=torch.long, device=dev)
        tc[:, :t] = content_target
        tcp = torch.full((b, tp), self.coarse_position_pad_code, dtype=torch.long, device=dev)
        tfp = torch.full((b, tp), self.fine_position_pad_code, dtype=torch.long, device=dev)
        if self.activate_pad_ignore:
            tcp[:, :lc - 1] = coarse_position_target
            tfp[:, lc - 1:t] = fine_position_target
        else:
            tcp[:, :lc] = coarse_position_target
            tfp[:, lc:t] = fine_position_target
        return (tc.view(-1), ign_c), (tcp.view(-1), self.coarse_position_pad_code), (tfp.view(-1), self.fine_position_pad_code)


class _StackGPTLossFn(torch.autograd.Function):
    """teacher-forced forward + the three cross entropies as one autograd node"""

    @staticmethod
    def forward(ctx, mod, want_grad, inputs, *params):
        (cc, fc, cp, fp, cs, fs, ct, cpt, fpt) = inputs
        ctx.mod, ctx.n = mod, len(params)
        tape = Tape() if want_grad else None
        with torch.no_grad():
            pl, cl, b, t, tp = mod.fwd(cc, fc, cp, fp, cs, fs, tape)
            dev = pl.device
            tg = mod._targets(b, t, tp, cp.shape[1], ct, cpt, fpt, dev)
            vp, vc = mod.config.fine_position_size, mod.config.vocab_size
            acc = torch.zeros(3, 2, dtype=torch.float32, device=dev)         # (loss_sum, count) per loss
            K.cross_entropy(cl, vc, tg[0][0], tg[0][1], acc[0, 0:1], acc[0, 1:2])
            K.cross_entropy(pl, vp, tg[1][0], tg[1][1], acc[1, 0:1], acc[1, 1:2])
            K.cross_entropy(pl, vp, tg[2][0], tg[2][1], acc[2, 0:1], acc[2, 1:2])
            losses = acc[:, 0] / acc[:, 1]
            content_loss, coarse_loss, fine_loss = losses[0], losses[1], losses[2]
            position_loss = (coarse_loss + fine_loss) / 2
        ctx.state = (tape, pl, cl, tg, acc, b, tp, vp, vc) if want_grad else None
        return position_loss, content_loss.clone(), coarse_loss.clone(), fine_loss.clone()

    @staticmethod
    def backward(ctx, g_pos, g_con, g_coarse, g_fine):
        if ctx.state is None:
            return (None,) * (3 + ctx.n)
        tape, pl, cl, tg, acc, b, tp, vp, vc = ctx.state
        mod = ctx.mod
        with torch.no_grad():
            z = torch.zeros((), device=pl.device)
            g_pos = z if g_pos is None else g_pos
            gc = ((z if g_con is None else g_con) / acc[0, 1]).float().reshape(1)
            gcp = ((g_pos * 0.5 + (z if g_coarse is None else g_coarse)) / acc[1, 1]).float().reshape(1)
            gfp = ((g_pos * 0.5 + (z if g_fine is None else g_fine)) / acc[2, 1]).float().reshape(1)
            scratch = torch.zeros(2, dtype=torch.float32, device=pl.device)
            d_cl = K.cross_entropy(cl, vc, tg[0][0], tg[0][1], scratch[0:1], scratch[1:2], gc.contiguous(), True)
            d_pl = K.cross_entropy(pl, vp, tg[1][0], tg[1][1], scratch[0:1], scratch[1:2], gcp.contiguous(), True)
            d_pl = K.add(d_pl, K.cross_entropy(pl, vp, tg[2][0], tg[2][1], scratch[0:1], scratch[1:2], gfp.contiguous(), True))
            mod.bwd(d_pl, d_cl, tape, b, tp)
        return (None,) * (3 + ctx.n)
