"""Minimal trainer honouring the reference's LightningModule surface, without pytorch_lightning.

Covers what train.py + Lightning 1.5.6 do for the DQ-VAE path (train.py:227-270):
  * two-optimizer automatic optimisation: for optimizer_idx in (0, 1): training_step -> backward -> step;
  * LambdaLR schedules stepped every batch (models/stage1/utils.py:6-24);
  * data parallelism: one process per GPU, gradients averaged with bucketed RCCL all-reduce launched on a
    side stream as soon as the backward has finished (HIP kernels write .grad in place, so buckets are
    filled by `GradBuckets.reduce`, not by autograd hooks);
  * HipAdam: torch.optim.Optimizer subclass whose step() is the fused dvq_adam kernel per tensor.
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.distributed as dist

from . import kernels as K
from . import runtime as rt


# ---- LR schedules (models/stage1/utils.py:6-24) ----------------------------------------------------
def _fn_linear_warmup(warmup_steps, step):
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return 1.0


def scheduler_linear_warmup(warmup_steps):
    return partial(_fn_linear_warmup, warmup_steps)


def _fn_linear_warmup_cosine_decay(warmup_steps, max_steps, multipler_min, step):
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    multipler = 0.5 * (math.cos((step - warmup_steps) / (max_steps - warmup_steps) * math.pi) + 1)
    return max(multipler, multipler_min)


def scheduler_linear_warmup_cosine_decay(warmup_steps, max_steps, multipler_min):
    return partial(_fn_linear_warmup_cosine_decay, warmup_steps, max_steps, multipler_min)


# ---- flat parameter / gradient storage -------------------------------------------------------------------
class FlatParams:
    """One flat fp32 buffer for the parameters and one for their gradients; every Parameter's .data / .grad is
    a view into them.  The HIP kernels accumulate gradients in place into the flat buffer, which is also the
    RCCL all-reduce buffer and the operand of ONE fused Adam launch (instead of one per tensor)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                new = self._view(self.flat_p, off, p)
                new.copy_(p.data)
                p.data = new
                p.grad = self._view(self.flat_g, off, p)
                off += n
        rt.bump_weights_epoch()      # parameter storage moved: packed copies must be rebuilt

    @staticmethod
    def _view(flat, off, p):
        """view of the flat buffer with p's logical shape; 4-D conv weights are stored channel-last
        ([Cout][KH][KW][Cin]): the wgrad kernel then accumulates with contiguous atomics and the bf16 forward
        pack is a plain cast"""
        n = p.numel()
        if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
            co, ci, kh, kw = p.shape
            return flat[off:off + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    def attach_grads(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self._view(self.flat_g, off, p)
            off += n

    def zero_grad(self):
        self.flat_g.zero_()
        self.attach_grads()


# ---- optimizer ------------------------------------------------------------------------------------------
class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics (no amsgrad; `weight_decay` is decoupled like AdamW, 0 = plain Adam).  With
    `flatten()` (done by the Trainer) all parameter groups live in ONE flat buffer pair, each group a contiguous segment
    updated by one fused kernel launch with the group's lr / weight decay; otherwise one launch per tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = None
        self._fstate = None

    def flatten(self) -> FlatParams:
        if self.flat is None:
            allp = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
            self.flat = FlatParams(allp)
            self._segments = []
            off = 0
            for g in self.param_groups:
                n = sum(p.numel() for p in g["params"] if p.requires_grad)
                self._segments.append((off, n))
                off += n
            for p in self.flat.params:
                p._dvq_group = id(self)
            self._fstate = {"step": 0, "m": torch.zeros_like(self.flat.flat_p), "v": torch.zeros_like(self.flat.flat_p)}
        return self.flat

    @torch.no_grad()
    def step(self, closure=None):
        if self.flat is not None:
            st = self._fstate
            st["step"] += 1
            for g, (off, n) in zip(self.param_groups, self._segments):
                if n == 0:
                    continue
                sl = slice(off, off + n)
                K.adamw_step(self.flat.flat_p[sl], self.flat.flat_g[sl], st["m"][sl], st["v"][sl], g["lr"], g["betas"][0],
                             g["betas"][1], g["eps"], g.get("weight_decay", 0.0), st["step"])
            rt.bump_group_epoch(id(self))      # only this optimizer's packed weights are stale
            return
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                K.adamw_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                             group.get("weight_decay", 0.0), st["step"])
        rt.bump_weights_epoch()


# ---- data-parallel gradient exchange --------------------------------------------------------------------
class GradBuckets:
    """All-reduce view of a FlatParams gradient buffer: fixed-size chunks, last layers first (their gradients are
    final first), launched asynchronously on RCCL's stream."""

    def __init__(self, params_or_flat, bucket_bytes=64 << 20, process_group=None):
        self.fp = params_or_flat if isinstance(params_or_flat, FlatParams) else FlatParams(params_or_flat)
        self.pg = process_group
        n = self.fp.flat_g.numel()
        step = max(1, bucket_bytes // 4)
        self.bucket_elems = step
        self.flat = [self.fp.flat_g[i:min(n, i + step)] for i in range(0, n, step)]
        self.params = self.fp.params
        self._pending, self._done = [], []

    def zero(self):
        self.fp.zero_grad()

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1

    def reduce_range(self, lo, hi):
        """start averaging flat_g[lo:hi] over ranks NOW (asynchronously, on RCCL's stream) -- called from inside the
        backward as soon as that part of the gradient is final, so the exchange overlaps the rest of the backward"""
        if not self._active() or hi <= lo:
            return
        ws = dist.get_world_size(self.pg)
        seg = self.fp.flat_g[lo:hi]
        seg.div_(ws)
        step = max(1, self.bucket_elems)
        for a in range(lo, hi, step):
            b = min(hi, a + step)
            self._pending.append(dist.all_reduce(self.fp.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._done.append((lo, hi))

    def reduce(self, async_op=True):
        """average the ranges that reduce_range() has not covered yet (no-op for world size 1); returns ALL work handles"""
        if not self._active():
            return []
        n = self.fp.flat_g.numel()
        covered = sorted(self._done)
        pos = 0
        gaps = []
        for lo, hi in covered:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            gaps.append((pos, n))
        for lo, hi in reversed(gaps):              # last layers first
            self.reduce_range(lo, hi)
        works, self._pending, self._done = [w for w in self._pending if w is not None], [], []
        return works

    def param_range(self, params):
        """[lo, hi) of the flat buffer covered by `params` (they must be contiguous in the optimizer's order)"""
        ids = {id(p) for p in params}
        off, lo, hi = 0, None, None
        for p in self.fp.params:
            if id(p) in ids:
                lo = off if lo is None else lo
                hi = off + p.numel()
            off += p.numel()
        return (lo or 0), (hi or 0)


class DataModuleFromConfig:
    """data/build.py:16-90 stand-in: BASELINE configs run on synthetic batches."""

    def __init__(self, batch_size, train=None, validation=None, test=None, wrap=False, num_workers=None, **kw):
        self.batch_size = batch_size
        self.cfg = dict(train=train, validation=validation, test=test)

    def prepare_data(self):
        pass


class Trainer:
    """fit loop for a DualGrainVQModel-like module on batches produced by `batch_fn(step) -> dict`."""

    def __init__(self, model, max_steps, log_every=0):
        self.model, self.max_steps, self.log_every = model, max_steps, log_every
        self.opts, self.scheds = model.configure_optimizers()
        self.buckets = [GradBuckets(o.flatten()) for o in self.opts]
        import inspect
        # Lightning passes optimizer_idx only to modules that declare it (two-optimizer stage 1); stage 2 has one optimizer
        self._takes_opt_idx = "optimizer_idx" in inspect.signature(model.training_step).parameters

    def train_step(self, batch, batch_idx):
        m = self.model
        losses = []
        K.arena_reset(self.buckets[0].fp.flat_g.device)
        for oi, opt in enumerate(self.opts):
            self.buckets[oi].zero()
            self._arm_overlap(oi)
            loss = m.training_step(batch, batch_idx, oi) if self._takes_opt_idx else m.training_step(batch, batch_idx)
            if loss.requires_grad:
                loss.backward()
            works = self.buckets[oi].reduce()
            for w in works:
                w.wait()
            opt.step()
            self.scheds[oi]["scheduler"].step()
            losses.append(loss.detach())
        m.global_step += 1
        return losses

    def _arm_overlap(self, oi):
        """autoencoder optimizer: the decoder-side gradients (decoder, quant convs) are final before the encoder's backward
        starts -- their all-reduce is launched from inside the backward (model._grad_hook) and overlaps the encoder backward"""
        m = self.model
        if not hasattr(m, "_grad_hook"):
            return
        m._grad_hook = None
        gb = self.buckets[oi]
        if oi == 0 and gb._active() and hasattr(m, "encoder"):
            lo, hi = gb.param_range([p for p in m.encoder.parameters() if p.requires_grad])
            n = gb.fp.flat_g.numel()
            if lo == 0 and 0 < hi < n:
                m._grad_hook = lambda tag: gb.reduce_range(hi, n) if tag == "decoder_side_done" else None

    # ---- checkpoint / resume (train.py:153-185, 270 + `-r`: Lightning's last.ckpt carries the model state_dict, the optimizer
    # and scheduler states and the global step; same top-level keys here so that `state_dict` stays loadable by the reference) ----
    def state_dict(self):
        opt_states = []
        for o in self.opts:
            st = o._fstate
            opt_states.append({"step": int(st["step"]), "exp_avg": st["m"].detach().cpu(), "exp_avg_sq": st["v"].detach().cpu()})
        return {"state_dict": self.model.state_dict(), "global_step": int(self.model.global_step),
                "optimizer_states": opt_states, "lr_schedulers": [s["scheduler"].state_dict() for s in self.scheds]}

    @torch.no_grad()
    def load_state_dict(self, ckpt, strict=True):
        """resume: parameters are written THROUGH the flat-buffer views (copy_), the packed compute-dtype copies are
        invalidated, Adam moments / step counts / LambdaLR positions restored when the checkpoint has them"""
        self.model.load_state_dict(ckpt["state_dict"], strict=strict)
        for b in self.buckets:
            b.fp.attach_grads()
        rt.bump_weights_epoch()
        for o, st in zip(self.opts, ckpt.get("optimizer_states", [])):
            o._fstate["step"] = int(st["step"])
            o._fstate["m"].copy_(st["exp_avg"].to(o._fstate["m"].device))
            o._fstate["v"].copy_(st["exp_avg_sq"].to(o._fstate["v"].device))
        for s, st in zip(self.scheds, ckpt.get("lr_schedulers", [])):
            s["scheduler"].load_state_dict(st)
            for g, lr in zip(s["scheduler"].optimizer.param_groups, s["scheduler"].get_last_lr()):
                g["lr"] = lr
        self.model.global_step = int(ckpt.get("global_step", 0))

    def fit(self, batch_fn):
        self.model.train()
        for step in range(int(self.model.global_step), self.max_steps):
            losses = self.train_step(batch_fn(step), step)
            if self.log_every and step % self.log_every == 0:
                print(f"step {step}: " + " ".join(f"{float(l):.5f}" for l in losses), flush=True)
