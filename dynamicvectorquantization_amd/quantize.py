"""Host-side mirror of /root/reference/modules/vector_quantization/quantize2_mask.py
(VQEmbedding :10-132, VectorQuantize2 :135-210) on the libdvq_hip VQ kernels.

Same constructor kwargs, buffers (`cluster_size_ema`, `embed_ema`), parameter (`weight` [K+1, D], row K =
padding row, requires_grad False under EMA) and return signatures.  Differences that matter:
  * the [N,K] distance matrix and the [K,N] one-hot matrix are never formed;
  * the argmin is the mathematically exact one (oracle/vq.py), lowest index on ties;
  * the two data-parallel all-reduces (:86-88) AND the restart broadcast (:99-100) are ONE all-reduce of a flat buffer
    ([K, D+1] statistics | [K, D] restart rows: rank 0's candidates + zeros from the other ranks) -- SURVEY.md section 5/8e;
  * RNG: the restart candidate rows are k distinct rows drawn by dvq_sample_rows (keyed Feistel permutation, state in
    device memory) unless `restart_perm` is injected (tests): the reference's randperm stream cannot be reproduced.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import kernels as K
from . import runtime as rt
from .layers import Tape, to_nchw, to_nhwc

_FORCE_DP = os.environ.get("DVQ_FORCE_DP", "0") == "1"     # exchange even in a one-rank group (single-GPU test of the DP path)


class VQEmbedding(nn.Embedding):
    """VQ embedding module with EMA update (quantize2_mask.py:10-132)."""

    def __init__(self, n_embed, embed_dim, ema=True, decay=0.99, restart_unused_codes=True, eps=1e-5):
        super().__init__(n_embed + 1, embed_dim, padding_idx=n_embed)
        self.ema, self.decay, self.eps = ema, decay, eps
        self.restart_unused_codes = restart_unused_codes
        self.n_embed = n_embed
        if self.ema:
            for p in self.parameters():
                p.requires_grad_(False)
            self.register_buffer("cluster_size_ema", torch.zeros(n_embed))
            self.register_buffer("embed_ema", self.weight[:-1, :].detach().clone())
        self._prep = None          # (version, prep buffer)
        self._cb_version = 0       # bumped by every in-place codebook rewrite (EMA); conv packs are unaffected
        self.restart_perm = None   # optional injected permutation (tests)
        self.track_flagged = False  # True: every search leaves its re-rank counters in `last_flagged`
        self.last_flagged = None   # device int32 [3]: rows settled in fp64 by the last search (all codes, candidate list, of those: class scans only)

    # -- search -------------------------------------------------------------------------------------
    def _codebook(self):
        return self.weight.detach()[:-1]    # contiguous leading rows

    def _prepared(self):
        ent = self._prep
        ver = (self._cb_version, rt.weights_epoch(), rt.codebook_epoch())
        if ent is None or ent[0] != ver or ent[1].device != self.weight.device:
            ent = (ver, K.vq_prepare(self._codebook()))
            self._prep = ent
        return ent[1]

    @torch.no_grad()
    def compute_distances(self, inputs):
        """inputs [..., D] -> fp32 [..., K] squared distances |x|^2 + |e|^2 - 2 x.e (quantize2_mask.py:29-48).  Analysis API:
        the search itself (find_nearest_embedding, forward) never forms this matrix; here it is what the caller asked for,
        written by dvq_vq_distances in row chunks."""
        d = inputs.shape[-1]
        assert d == self.weight.shape[-1]
        flat = inputs.reshape(-1, d)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        if flat.dtype not in (torch.float32, torch.bfloat16):
            flat = flat.float()
        return K.vq_distances(flat, self._codebook()).reshape(*inputs.shape[:-1], -1)

    @torch.no_grad()
    def find_nearest_embedding(self, inputs):
        """inputs [..., D] -> int64 [...] exact nearest code (quantize2_mask.py:50-55)."""
        d = inputs.shape[-1]
        flat = inputs.reshape(-1, d)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        cb = self._codebook()
        prep = self._prepared() if d in (64, 128, 256) else None
        if self.track_flagged:          # diagnostics: a private copy of the search's re-rank counters (one extra small kernel)
            idx, self.last_flagged = K.vq_argmin(flat, cb, prep, impl=rt.impl(), return_flagged=True)
        else:
            idx = K.vq_argmin(flat, cb, prep, impl=rt.impl())
        return idx.reshape(inputs.shape[:-1])

    # -- EMA ------------------------------------------------------------------------------------------
    def _rng(self, device):
        """device-resident {seed, counter} of the restart-row sampler (seeded from torch's seed; the counter advances on the
        stream, so the draw needs no host state and replays inside a captured training step)"""
        st = getattr(self, "_rng_state", None)
        if st is None or st.device != device:
            st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64).to(device)
            self._rng_state = st
        return st

    @torch.no_grad()
    def _update_buffers(self, vectors, idxs):
        """quantize2_mask.py:66-105.  vectors [N,D] (compute dtype), idxs [N]."""
        k, d = self.n_embed, self.weight.shape[-1]
        vectors = vectors.reshape(-1, d)
        idxs = idxs.reshape(-1)
        # the exchanged buffer is persistent (one per module): stable addresses for RCCL and for a recorded step.  ONE flat fp32
        # buffer = [K, D+1] statistics (sums | count) followed by the [K, D] restart candidate rows, so that the data-parallel
        # exchange is a single collective over it (see _exchange)
        xb = getattr(self, "_xchg", None)
        if xb is None or xb[0].device != vectors.device:
            xb = self._xchg = self._exchange_buffers(k, d, vectors.device)
        stats = K.vq_ema_stats(vectors, idxs, k, out=xb[1])           # [K, D+1] = (sums | count)
        restart = None
        dp = self._dp_active()
        if self.restart_unused_codes:
            if dp and dist.get_rank() != 0:
                # only rank 0's candidate rows are used (quantize2_mask.py:99-100 broadcasts them): the others contribute
                # exact zeros to the fused all-reduce and draw nothing
                restart = xb[2].zero_()
            else:
                n = vectors.shape[0]
                src = K.cast(vectors, torch.float32)
                if n < k:
                    # _tile_with_noise (quantize2_mask.py:57-64): repeat rows and add U(0,1) * 0.01/sqrt(D);
                    # only reachable with tiny batches; the noise comes from the module's device generator (replay-safe)
                    reps = (k + n - 1) // n
                    src = K.add_uniform_(src.repeat(reps, 1), 0.01 / np.sqrt(d), self._rng(vectors.device))
                    n = src.shape[0]
                if self.restart_perm is not None:
                    perm = self.restart_perm.to(vectors.device)[:k]
                else:
                    perm = K.sample_rows(k, n, self._rng(vectors.device))     # = randperm(n)[:k]: k distinct rows
                restart = K.vq_embed(src.contiguous(), perm, out=xb[2])
        if dp:
            self._exchange(xb[0] if restart is not None else stats)
        K.vq_ema_apply(stats, restart, self.decay, self.eps, self.cluster_size_ema, self.embed_ema, self.weight.data)
        self._cb_version += 1

    @staticmethod
    def _exchange_buffers(k, d, device):
        """(flat, statistics view [K, D+1], restart-rows view [K, D]) of one fp32 allocation"""
        flat = torch.empty(k * (d + 1) + k * d, dtype=torch.float32, device=device)
        return flat, flat[: k * (d + 1)].view(k, d + 1), flat[k * (d + 1):].view(k, d)

    @staticmethod
    def _dp_active():
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_DP)

    @staticmethod
    def _exchange(buf):
        """Data-parallel step of the EMA update as ONE collective: all-reduce(SUM) of the flat buffer holding the [K, D+1]
        statistics and, behind them, the restart candidate rows -- rank 0's rows plus exact zeros from every other rank, i.e. the
        reference's `broadcast(_vectors_random, 0)` (quantize2_mask.py:99-100) folded into its two all-reduces (:86-88): one
        launch and one latency per VQ forward instead of three.  (Sending only the rows of DEAD codes would need the dead set
        on the host before the collective is sized -- a device-to-host sync in the middle of every training forward, which a
        recorded step cannot have; the whole [K, D] block is 1 MB at K = 1024.)  Device-agnostic (tested on CPU tensors over gloo)."""
        def exchange():          # eager even inside a captured training step (runtime.graph_break)
            if os.environ.get("DVQ_DP_NOOP_COLLECTIVES", "0") == "1":
                return
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        rt.graph_break(exchange)

    def forward(self, inputs):
        """inputs [B, N, D] -> (embeds, idx) with the reference's ordering (quantize2_mask.py:117-128):
        search, EMA buffer update, gather with the OLD weight, then rewrite the weight."""
        idx = self.find_nearest_embedding(inputs)
        embeds = self.embed(idx)
        if self.training and self.ema:
            self._update_buffers(inputs, idx)
        return embeds, idx

    def embed(self, idxs):
        return K.vq_embed(self.weight.detach(), idxs.contiguous(), torch.float32)


class VectorQuantize2(nn.Module):
    """quantize2_mask.py:135-210."""

    def __init__(self, codebook_size, codebook_dim=None, accept_image_fmap=True, commitment_beta=0.25, decay=0.99,
                 restart_unused_codes=True, channel_last=False):
        super().__init__()
        self.accept_image_fmap = accept_image_fmap
        self.beta = commitment_beta
        self.channel_last = channel_last
        self.restart_unused_codes = restart_unused_codes
        self.codebook = VQEmbedding(codebook_size, codebook_dim, decay=decay, restart_unused_codes=restart_unused_codes)
        self.codebook.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)
        if not accept_image_fmap:
            raise NotImplementedError("only accept_image_fmap=True is on the shipped configs' path")

    # -- NHWC core used by the model -----------------------------------------------------------------
    def fwd(self, h, mask, tape):
        """h NHWC [B,H,W,D] (compute dtype), mask fp32 [B,H,W] or None ->
        (x_q NHWC, loss fp32 scalar tensor, idx int64 [B,H,W])"""
        b, hh, ww, d = h.shape
        flat = h.view(-1, d)
        cbk = self.codebook
        idx = cbk.find_nearest_embedding(flat)
        mflat = None if mask is None else mask.reshape(-1)
        # gather with the OLD weight before the EMA rewrite (quantize2_mask.py:123-126)
        xq, loss_sum = K.vq_gather_loss(flat, cbk._codebook(), idx, mflat)
        if cbk.training and cbk.ema:
            if tape is not None:
                # a kernel copy: torch's clone() of a contiguous tensor is a hipMemcpy node in a recorded step, and launch lists
                # (csrc/cmdlist.hip) re-issue kernel nodes only.  x * 1 is exact for every value
                tape.s["cb_old"] = cbk._codebook().mul(1)
            cbk._update_buffers(flat, idx)
        n_el = flat.numel()
        loss = (loss_sum * ((1.0 + self.beta) / n_el)).to(torch.float32).reshape(())
        if tape is not None:
            tape.s.update(h=flat, idx=idx, mask=mflat, n_el=n_el)
            tape.s.setdefault("cb_old", cbk._codebook())
        return xq.view(b, hh, ww, d), loss, idx.view(b, hh, ww)

    def bwd(self, g_xq, g_loss, tape):
        """g_xq NHWC grad of x_q, g_loss device scalar (grad of the loss) -> grad of h (NHWC)"""
        s = tape.s
        d = s["h"].shape[1]
        coef = (g_loss.to(torch.float32) * (2.0 * self.beta / s["n_el"])).reshape(1).contiguous()
        dx = K.vq_backward(g_xq.reshape(-1, d), s["h"], s["cb_old"], s["idx"], s["mask"], coef)
        return dx.view(g_xq.shape)

    # -- reference signature ---------------------------------------------------------------------------
    def forward(self, x, codebook_mask=None, *ignorewargs, **ignorekwargs):
        """x [B,D,H,W] -> (x_q [B,D,H,W], loss, (None, None, idx [B,H,W]))  (quantize2_mask.py:157-191)"""
        mask = None
        if codebook_mask is not None:
            mask = codebook_mask.reshape(codebook_mask.shape[0], *codebook_mask.shape[-2:]).to(torch.float32).contiguous()
        xq, loss, idx = _VQFn.apply(self, x, mask)
        return xq, loss, (None, None, idx)

    @torch.no_grad()
    def get_soft_codes(self, x, temp=1.0, stochastic=False):
        """quantize2_mask.py:193-205: soft_code = softmax(-distances / temp); code = a multinomial draw (device RNG) or the
        nearest code.  x [..., D] like the reference (it passes x straight to compute_distances)."""
        distances = self.codebook.compute_distances(x)
        k = distances.shape[-1]
        flat = distances.reshape(-1, k)
        soft = K.softmax_rows(flat, flat.shape[0], k, -1.0 / float(temp)).reshape(distances.shape)
        if stochastic:
            code = torch.multinomial(soft.reshape(-1, k), 1).reshape(*soft.shape[:-1])
        else:
            code = self.codebook.find_nearest_embedding(x)      # the exact nearest code (= argmin of the distances up to fp32 ties)
        return soft, code

    def get_codebook_entry(self, indices, *kwargs):
        return self.codebook.embed(indices)   # (batch, height, width, channel)


class _VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, mask):
        ctx.module, ctx.in_dtype = module, x.dtype
        ctx.tape = Tape()
        with torch.no_grad():
            h = to_nhwc(x, rt.compute_dtype())
            xq, loss, idx = module.fwd(h, mask, ctx.tape)
            xq = to_nchw(K.cast(xq, x.dtype))
        ctx.mark_non_differentiable(idx)
        return xq, loss, idx

    @staticmethod
    def backward(ctx, g_xq, g_loss, _g_idx):
        with torch.no_grad():
            if g_loss is None:
                g_loss = torch.zeros((), device=g_xq.device)
            dx = ctx.module.bwd(to_nhwc(g_xq, rt.compute_dtype()), g_loss, ctx.tape)
            dx = to_nchw(K.cast(dx, ctx.in_dtype))
        return None, dx, None
