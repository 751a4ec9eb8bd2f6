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


# ---- optimizer ---------------------------------------------------------------------------------------
class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay / amsgrad), one fused HIP kernel per parameter tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
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
                K.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"], st["step"])
        rt.bump_weights_epoch()


# ---- data-parallel gradient exchange --------------------------------------------------------------------
class GradBuckets:
    """Flat fp32 buckets over a parameter list; .grad of every parameter is a view into its bucket, so the
    kernels accumulate straight into communication buffers and no copy is needed before the all-reduce."""

    def __init__(self, params, bucket_bytes=64 << 20, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.pg = process_group
        self.buckets = []
        cur, cur_n = [], 0
        for p in self.params:
            if cur and (cur_n + p.numel()) * 4 > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat = []
        for b in self.buckets:
            n = sum(p.numel() for p in b)
            flat = torch.zeros(n, dtype=torch.float32, device=b[0].device)
            off = 0
            for p in b:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.flat.append(flat)

    def zero(self):
        for f in self.flat:
            f.zero_()

    def reduce(self, async_op=True):
        """average gradients over ranks (no-op for world size 1); returns work handles"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.pg) == 1:
            return []
        ws = dist.get_world_size(self.pg)
        works = []
        for f in reversed(self.flat):   # backward fills the last layers' buckets first
            f.div_(ws)
            works.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op))
        return [w for w in works if w is not None]


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
        self.buckets = [GradBuckets(sum((g["params"] for g in o.param_groups), [])) for o in self.opts]

    def train_step(self, batch, batch_idx):
        m = self.model
        losses = []
        for oi, opt in enumerate(self.opts):
            self.buckets[oi].zero()
            loss = m.training_step(batch, batch_idx, oi)
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

    def fit(self, batch_fn):
        self.model.train()
        for step in range(self.max_steps):
            losses = self.train_step(batch_fn(step), step)
            if self.log_every and step % self.log_every == 0:
                print(f"step {step}: " + " ".join(f"{float(l):.5f}" for l in losses), flush=True)
