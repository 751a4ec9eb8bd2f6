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

import os

import torch
import torch.distributed as dist

from . import kernels as K
from . import runtime as rt

_FORCE_DP = os.environ.get("DVQ_FORCE_DP", "0") == "1"
_NOOP_COLL = os.environ.get("DVQ_DP_NOOP_COLLECTIVES", "0") == "1"      # debugging: exchange points without the RCCL calls
_NO_HOOK = os.environ.get("DVQ_DP_NO_HOOK", "0") == "1"                 # debugging: no all-reduce launch from inside the backward


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
    # (exactly the reference's expression, pinned bit for bit by tests/golden/losses.npz incl. steps past `max_steps`, where it climbs
    #  again like the reference's does: train.py keeps that from happening by sizing `training_steps` from the real loader BEFORE
    #  the schedules are built)
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
            # per-group hyper-parameters in DEVICE memory (dvq_adamw_dev): the launch has no step-dependent argument, so a
            # captured training step follows the LR schedule -- the host rewrites this table before each step
            self._hyper = torch.zeros(len(self.param_groups), 8, dtype=torch.float32, device=self.flat.flat_p.device)
            self._prepared = False
        return self.flat

    @torch.no_grad()
    def prepare_step(self):
        """upload the hyper-parameters of the NEXT step() (lr from the schedule, bias corrections of step count + 1).  Called
        by the Trainer at the top of every step, outside any capture; step() calls it itself when nobody did."""
        t = self._fstate["step"] + 1
        for gi, g in enumerate(self.param_groups):
            b1, b2 = g["betas"]
            lr = float(g["lr"])
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            K.set_f32x8(self._hyper[gi], [lr / bc1, b1, b2, g["eps"], 1.0 / math.sqrt(bc2), 1.0 - lr * g.get("weight_decay", 0.0)])
        self._prepared = True

    @torch.no_grad()
    def step(self, closure=None):
        if self.flat is not None:
            st = self._fstate
            if not self._prepared:
                self.prepare_step()
            self._prepared = False
            st["step"] += 1
            for gi, (off, n) in enumerate(self._segments):
                if n == 0:
                    continue
                sl = slice(off, off + n)
                K.adamw_dev(self.flat.flat_p[sl], self.flat.flat_g[sl], st["m"][sl], st["v"][sl], self._hyper[gi])
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
        self.launched = 0                 # all-reduce launches so far (tests: every bucket exactly once per step)
        self._divisor = None              # tests: pre-division factor of a pretended world size in a one-rank group
        self.measure = None               # list of (start, end) event pairs around wait() while bench.py measures the exposed time

    def zero(self):
        self.fp.zero_grad()

    def _active(self):
        # DVQ_FORCE_DP=1: exchange even in a one-rank group (exercises the RCCL / capture-break path on a single GPU)
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.pg) > 1 or _FORCE_DP)

    def reduce_range(self, lo, hi):
        """start averaging flat_g[lo:hi] over ranks NOW (asynchronously, on RCCL's stream) -- called from inside the
        backward as soon as that part of the gradient is final, so the exchange overlaps the rest of the backward"""
        if not self._active() or hi <= lo:
            return
        ws = self._divisor if self._divisor is not None else dist.get_world_size(self.pg)
        # weight gradients of eager steps are accumulated into this range by kernels on the side stream (runtime.side_wgrad):
        # the pre-division below reads the range on the CURRENT stream, so the side stream is joined first -- not only at the
        # graph break further down
        rt.join_side()
        seg = self.fp.flat_g[lo:hi]
        seg.div_(ws)
        step = max(1, self.bucket_elems)
        chunks = [self.fp.flat_g[a:min(hi, a + step)] for a in range(lo, hi, step)]

        def launch():        # eager even when the step is being captured (rt.graph_break): RCCL runs on its own stream
            if _NOOP_COLL:
                self.launched += len(chunks)
                return
            for c in chunks:
                self._pending.append(dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            self.launched += len(chunks)
        rt.graph_break(launch)
        self._done.append((lo, hi))

    def reduce_params(self, params, min_bytes=1 << 20):
        """gradients of `params` are final: start the all-reduce of the contiguous flat ranges they cover (called from inside the
        backward, per block / level / segment, so that the exchange overlaps the rest of the backward).  Runs shorter than
        `min_bytes` are left to the closing reduce() -- a latency-bound collective per bias vector helps nobody."""
        if not self._active():
            return
        offs = self._offsets()
        spans = sorted((offs[id(p)], offs[id(p)] + p.numel()) for p in params if id(p) in offs)
        runs = []
        for lo, hi in spans:
            if runs and lo <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], hi)
            else:
                runs.append([lo, hi])
        for lo, hi in runs:
            if (hi - lo) * 4 < min_bytes or any(lo < d_hi and d_lo < hi for d_lo, d_hi in self._done):
                continue
            self.reduce_range(lo, hi)

    def _offsets(self):
        offs = getattr(self, "_offs", None)
        if offs is None:
            offs, off = {}, 0
            for p in self.fp.params:
                offs[id(p)] = off
                off += p.numel()
            self._offs = offs
        return offs

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
        self._done = []
        return list(self._pending)

    def wait(self):
        """make the current stream wait for every exchange launched so far (eager; a graph break inside a capture)"""
        if not self._active():
            return

        def _wait():
            if self.measure is not None:              # bench.py: time the compute stream spends in this wait (HIP events around it)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in self._pending:
                if w is not None:
                    w.wait()
            self._pending.clear()
            if self.measure is not None:
                e1.record()
                self.measure.append((e0, e1))
        rt.graph_break(_wait)

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


def reference_learning_rate(model_cfg, world, batch_size):
    """train.py:248-257 of the reference: lr = ngpu * batch_size * base_learning_rate when the config gives a base rate, else
    the absolute `learning_rate`"""
    if "base_learning_rate" in model_cfg:
        return world * batch_size * model_cfg["base_learning_rate"]
    if "learning_rate" in model_cfg:
        return model_cfg["learning_rate"]
    raise NotImplementedError("Please set learning rate!")


def _replicated_tensors(model):
    """(name, tensor) of everything that must be IDENTICAL on all ranks: parameters and buffers, except the statistics of
    the discriminator's BatchNorm layers, which DDP keeps per rank (modules/discriminator/model.py:31)"""
    from .layers import BatchNorm2d
    skip = set()
    for mn, mod in model.named_modules():
        if isinstance(mod, BatchNorm2d):
            skip.update(f"{mn}.{bn}" for bn, _ in mod.named_buffers(recurse=False))
    for n, p in model.named_parameters():
        yield n, p.detach()
    for n, b in model.named_buffers():
        if n not in skip:
            yield n, b


def broadcast_model(model, src=0):
    """DDP's initial synchronisation: rank `src`'s parameters and buffers everywhere (one broadcast per tensor at start-up)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return
    for _, t in _replicated_tensors(model):
        dist.broadcast(t, src)
    from .layers import ActNorm
    for m in model.modules():
        if isinstance(m, ActNorm):
            m._inited = None               # the `initialized` buffer was just overwritten in place: re-read it, do not trust the host mirror
    rt.bump_weights_epoch()


def replicas_equal(model):
    """debug check (DVQ_DP_CHECK_EVERY=k): True iff every replicated tensor has the same checksum on all ranks"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return True
    sums = torch.stack([t.double().sum() + t.double().abs().sum() * 3.0 for _, t in _replicated_tensors(model)])
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


class Trainer:
    """fit loop for a DualGrainVQModel-like module on batches produced by `batch_fn(step) -> dict`.

    Step capture: after `graph_after` eager steps with an unchanged signature (batch shapes, train flag, loss phase, runtime
    switches) the step is recorded once as hipGraph segments (runtime.StepGraph) and replayed from then on; collectives
    stay eager between segments, the optimizer reads its hyper-parameters from device memory, batches are copied into
    static input buffers.  `DVQ_STEP_GRAPH=0` (or use_graph=False) keeps every step eager."""

    def __init__(self, model, max_steps, log_every=0, use_graph=None, graph_after=3):
        self.model, self.max_steps, self.log_every = model, max_steps, log_every
        self.opts, self.scheds = model.configure_optimizers()
        self.buckets = [GradBuckets(o.flatten()) for o in self.opts]
        broadcast_model(model)             # identical start on every rank (DDP's parameter / buffer broadcast), not just equal seeds
        self.check_every = int(os.environ.get("DVQ_DP_CHECK_EVERY", "0"))
        import inspect
        # Lightning passes optimizer_idx only to modules that declare it (two-optimizer stage 1); stage 2 has one optimizer
        self._takes_opt_idx = "optimizer_idx" in inspect.signature(model.training_step).parameters
        if use_graph is None:
            use_graph = os.environ.get("DVQ_STEP_GRAPH", "1") != "0"
        self.use_graph = bool(use_graph) and bool(getattr(model, "GRAPH_SAFE", False))
        self.graph_after = graph_after
        self._graph = None            # dict(sg, sig, static, losses)
        self._last_sig, self._stable = None, 0
        self.graph_replays = 0

    # ---- one step ------------------------------------------------------------------------------------------------
    def train_step(self, batch, batch_idx):
        out = self._train_step(batch, batch_idx)
        if self.check_every and (int(self.model.global_step) % self.check_every) == 0 and not replicas_equal(self.model):
            raise RuntimeError(f"data-parallel replicas diverged at global step {self.model.global_step}")
        return out

    def _train_step(self, batch, batch_idx):
        if not self.use_graph:
            return self._eager_step(batch, batch_idx)
        sig = self._signature(batch)
        if sig is None:                            # profiled step / injected host-side noise: eager, the recording stays valid
            return self._eager_step(batch, batch_idx)
        g = self._graph
        if g is not None and g["sig"] == sig:
            return self._replay(batch)
        if g is not None:
            self._graph = None                     # the recorded control flow no longer applies
        self._stable = self._stable + 1 if sig == self._last_sig else 0
        self._last_sig = sig
        if self._stable >= self.graph_after:       # `graph_after` eager steps with this signature have run
            if self._capture(batch, batch_idx, sig):
                return self._replay(batch)
        return self._eager_step(batch, batch_idx)

    def _eager_step(self, batch, batch_idx):
        for o in self.opts:
            o.prepare_step()                       # hyper-parameters of this step -> device memory
        return self._step_body(batch, batch_idx)

    def _step_body(self, batch, batch_idx):
        """the capturable part: every launch of the two-optimizer step (collectives are graph breaks)"""
        m = self.model
        losses = []
        K.arena_reset(self.buckets[0].fp.flat_g.device)
        # single-threaded autograd: the backward's launches come from this thread (one capture thread, no thread hops)
        with torch.autograd.set_multithreading_enabled(False):
            for oi, opt in enumerate(self.opts):
                self.buckets[oi].zero()
                self._arm_overlap(oi)
                with rt.side_wgrad():          # conv weight gradients on a second stream, joined on exit (and at graph breaks)
                    loss = m.training_step(batch, batch_idx, oi) if self._takes_opt_idx else m.training_step(batch, batch_idx)
                    if loss.requires_grad:
                        loss.backward()
                self.buckets[oi].reduce()
                self.buckets[oi].wait()
                opt.step()
                self.scheds[oi]["scheduler"].step()
                losses.append(loss.detach())
        m.global_step += 1
        return losses

    # ---- capture / replay ----------------------------------------------------------------------------------------
    def _signature(self, batch):
        """everything host-side that shapes the step's launch sequence; None = this step cannot be captured"""
        m = self.model
        if K.profiling_active() or rt.capturing():
            return None
        extra = m.graph_signature() if hasattr(m, "graph_signature") else ()
        if extra is None:
            return None
        items = []
        for k in sorted(batch):
            v = batch[k]
            if torch.is_tensor(v):
                if not v.is_cuda:
                    return None
                items.append((k, tuple(v.shape), v.dtype, v.device.index))
            else:
                items.append((k, repr(v)))
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        return (tuple(items), bool(m.training), rt.compute_dtype(), rt.impl(), rt.fuse_gn_prologue(), world, extra)

    def _py_state(self):
        return ([o._fstate["step"] for o in self.opts], [s["scheduler"].state_dict() for s in self.scheds],
                [[g["lr"] for g in o.param_groups] for o in self.opts], int(self.model.global_step))

    def _set_py_state(self, st):
        steps, scheds, lrs, gstep = st
        for o, t, lr in zip(self.opts, steps, lrs):
            o._fstate["step"] = t
            for g, v in zip(o.param_groups, lr):
                g["lr"] = v
        for s, sd in zip(self.scheds, scheds):
            s["scheduler"].load_state_dict(sd)
        self.model.global_step = gstep

    def _capture(self, batch, batch_idx, sig) -> bool:
        """record one step (a dry run: captured kernels do not execute, the Python-side counters are put back)"""
        dev = self.buckets[0].fp.flat_g.device
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        saved = self._py_state()
        logged = dict(getattr(self.model, "_logged", {}))
        for o in self.opts:
            o.prepare_step()
        sg = rt.StepGraph(dev)
        try:
            with sg.capture():
                losses = self._step_body(static, batch_idx)
        except Exception as e:          # a launch sequence that cannot be captured: stay eager, say why once
            import warnings
            warnings.warn(f"training-step capture failed ({type(e).__name__}: {e}); continuing with eager launches")
            self.use_graph = False
            self._set_py_state(saved)
            for o in self.opts:
                o._prepared = False
            return False
        self._set_py_state(saved)
        if hasattr(self.model, "_logged"):
            # the step's log tensors now live in the graph's pool and are refreshed by every replay
            self.model._logged = {**logged, **self.model._logged}
        self._graph = {"sg": sg, "sig": sig, "static": static, "losses": losses}
        return True

    def _replay(self, batch):
        g = self._graph
        for k, v in batch.items():
            if torch.is_tensor(v) and v.data_ptr() != g["static"][k].data_ptr():
                g["static"][k].copy_(v, non_blocking=True)
        for o in self.opts:
            o.prepare_step()
        g["sg"].replay()
        # host-side bookkeeping the recorded launches do not carry
        for o, sc in zip(self.opts, self.scheds):
            o._fstate["step"] += 1
            o._prepared = False
            sc["scheduler"].step()
            # the replay rewrote this optimizer's parameters: whatever an EAGER forward (eval / encode / validation_step) packed
            # or prepared from them before is stale -- same invalidation an eager opt.step() / EMA update does
            rt.bump_group_epoch(id(o))
        rt.bump_codebook_epoch()
        self.model.global_step += 1
        self.graph_replays += 1
        return g["losses"]

    def drop_graph(self):
        self._graph, self._stable, self._last_sig = None, 0, None

    def _arm_overlap(self, oi):
        """Gradient exchange overlapped with the backward: every module of the model that declares `_grad_hook` calls it with the
        parameters whose gradients just became final (decoder side / encoder heads / each encoder level of the DQ-VAE, each
        block of StackGPT); under data parallelism that starts the all-reduce of their flat ranges right away
        (GradBuckets.reduce_params), the closing reduce() covers what is left.  Only the optimizer whose backward is about to
        run is armed."""
        gb = self.buckets[oi]
        armed = gb._active() and not _NO_HOOK
        for mod in self.model.modules():
            if hasattr(mod, "_grad_hook"):
                mod._grad_hook = gb.reduce_params if armed else None

    # ---- checkpoint / resume (train.py:153-185, 270 + `-r`) --------------------------------------------------------------------
    # Lightning's last.ckpt layout: {"state_dict", "optimizer_states": [torch Optimizer.state_dict()...], "lr_schedulers",
    # "global_step", "epoch"}.  The optimizer states are written in torch's own format (per-parameter exp_avg / exp_avg_sq in
    # the parameter's logical shape, indices over the optimizer's full parameter list, `param_groups`), so the file resumes
    # here AND under the reference; the round-1 private format ({"step","exp_avg","exp_avg_sq"} over the flat buffer) still loads.
    def _optimizer_state_dict(self, o):
        st, flat = o._fstate, o.flat
        offs, off = {}, 0
        for p in flat.params:
            offs[id(p)] = off
            off += p.numel()
        state, groups, idx = {}, [], 0
        for g in o.param_groups:
            ids = []
            for p in g["params"]:
                if id(p) in offs and st["step"] > 0:
                    state[idx] = {"step": torch.tensor(float(st["step"])),
                                  "exp_avg": FlatParams._view(st["m"], offs[id(p)], p).detach().cpu().contiguous().clone(),
                                  "exp_avg_sq": FlatParams._view(st["v"], offs[id(p)], p).detach().cpu().contiguous().clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def _load_optimizer_state(self, o, sd):
        st, flat = o._fstate, o.flat
        if "state" not in sd:                             # round-1 private format: flat buffers
            n = st["m"].numel()
            if sd["exp_avg"].numel() != n or sd["exp_avg_sq"].numel() != n:
                raise ValueError(f"optimizer state holds {sd['exp_avg'].numel()} elements, this optimizer has {n} "
                                 "(a different parameter set, e.g. disc_factor=0 dropping the discriminator?)")
            st["step"] = int(sd["step"])
            st["m"].copy_(sd["exp_avg"].to(st["m"].device))
            st["v"].copy_(sd["exp_avg_sq"].to(st["v"].device))
            return
        offs, off = {}, 0
        for p in flat.params:
            offs[id(p)] = off
            off += p.numel()
        full = [p for g in o.param_groups for p in g["params"]]
        n_saved = sum(len(g["params"]) for g in sd.get("param_groups", []))
        if n_saved and n_saved != len(full):
            raise ValueError(f"optimizer state lists {n_saved} parameters, this optimizer has {len(full)}")
        step = 0
        st["m"].zero_()
        st["v"].zero_()
        for idx, p in enumerate(full):
            ps = sd["state"].get(idx, sd["state"].get(str(idx)))
            if ps is None or id(p) not in offs:
                continue
            if tuple(ps["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {idx}: shape {tuple(ps['exp_avg'].shape)} != {tuple(p.shape)}")
            FlatParams._view(st["m"], offs[id(p)], p).copy_(ps["exp_avg"].to(st["m"].device))
            FlatParams._view(st["v"], offs[id(p)], p).copy_(ps["exp_avg_sq"].to(st["v"].device))
            step = max(step, int(float(ps["step"])))
        st["step"] = step

    def state_dict(self):
        return {"state_dict": self.model.state_dict(), "global_step": int(self.model.global_step),
                "epoch": int(getattr(self.model, "current_epoch", 0)),
                "optimizer_states": [self._optimizer_state_dict(o) for o in self.opts],
                "lr_schedulers": [s["scheduler"].state_dict() for s in self.scheds]}

    @torch.no_grad()
    def load_state_dict(self, ckpt, strict=True):
        """resume: parameters are written THROUGH the flat-buffer views (copy_), the packed compute-dtype copies are
        invalidated, Adam moments / step counts / LambdaLR positions restored when the checkpoint has them (torch / Lightning
        optimizer-state format or the round-1 flat format; a checkpoint whose optimizers do not match raises)"""
        self.model.load_state_dict(ckpt["state_dict"], strict=strict)
        for b in self.buckets:
            b.fp.attach_grads()
        rt.bump_weights_epoch()
        saved = ckpt.get("optimizer_states", [])
        if saved and len(saved) != len(self.opts):
            import warnings
            warnings.warn(f"checkpoint has {len(saved)} optimizer states, the model configures {len(self.opts)}: "
                          "optimizer moments are NOT restored (weights are)")
            saved = []
        for o, st in zip(self.opts, saved):
            self._load_optimizer_state(o, st)
            o._prepared = False
        for s, st in zip(self.scheds, ckpt.get("lr_schedulers", []) if saved else []):
            s["scheduler"].load_state_dict(st)
            for g, lr in zip(s["scheduler"].optimizer.param_groups, s["scheduler"].get_last_lr()):
                g["lr"] = lr
        self.model.global_step = int(ckpt.get("global_step", 0))

    def save_checkpoint(self, path):
        """atomic write (temp file + rename): a crash while saving never destroys the previous last.ckpt"""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = path + ".tmp"
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, path)

    @torch.no_grad()
    def validate(self, batches):
        """Lightning's validation loop (train.py:233-262 runs it every `check_val_every_n_epoch`): eval mode, `validation_step` on every
        batch, the logged scalars averaged over the batches -- and over the ranks under data parallelism (`sync_dist=True` of
        dqvae_dual_entropy.py:185-201) -- train mode restored.  -> {name: float}"""
        m = self.model
        was_training = m.training
        m.eval()
        sums, n = {}, 0
        try:
            for i, b in enumerate(batches):
                m._logged = {}
                m.validation_step(b, i)
                for k, v in m._logged.items():
                    if torch.is_tensor(v) and v.numel() == 1 or isinstance(v, (int, float)):
                        sums[k] = sums.get(k, 0.0) + (v.detach().double() if torch.is_tensor(v) else float(v))
                n += 1
        finally:
            m.train(was_training)
            m._logged = {}
        if n == 0:
            return {}
        keys = sorted(sums)
        dev = next(m.parameters()).device
        vec = torch.stack([torch.as_tensor(sums[k], dtype=torch.float64, device=dev).reshape(()) for k in keys]) / n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(vec)
            vec /= dist.get_world_size()
        return {k: float(x) for k, x in zip(keys, vec.cpu())}

    def fit(self, batch_fn, ckpt_path=None, save_every=0, is_rank0=True, val_fn=None, val_every=0, save_top_k=0):
        """`ckpt_path` + `save_every` (steps): rank 0 rewrites last.ckpt periodically, so a preempted run resumes with `-r`.
        `val_fn()` -> iterable of validation batches, run every `val_every` steps and after the last one; with a `monitor` on the model
        (the YAMLs say `monitor: val_rec_loss`) the `save_top_k` best checkpoints by that metric are kept next to last.ckpt as
        `epoch=<e>-<monitor>=<value>.ckpt` -- pytorch_lightning.callbacks.ModelCheckpoint(monitor, save_top_k, save_last=True, mode="min")
        of the reference's train.py:152-183"""
        self.model.train()
        best = []                                     # (value, path), ascending
        monitor = getattr(self.model, "monitor", None)
        if ckpt_path and is_rank0 and save_top_k and monitor:
            # a resumed run (-r) starts from the monitored checkpoints already on disk (Lightning keeps best_k_models inside the
            # checkpoint; here the file names carry the value): without this, old files were never compared or pruned again
            import glob
            import re
            pat = re.compile(r"epoch=\d+-" + re.escape(monitor) + r"=(-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?)\.ckpt$")
            for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(ckpt_path)), f"epoch=*-{glob.escape(monitor)}=*.ckpt")):
                mt = pat.search(os.path.basename(f))
                if mt:
                    best.append((float(mt.group(1)), f))
            best.sort(key=lambda t: t[0])
            for _, old in best[save_top_k:]:
                os.remove(old)
            del best[save_top_k:]

        def run_validation(step):
            metrics = self.validate(val_fn())
            self.last_val_metrics = metrics
            if is_rank0 and metrics:
                print(f"validation @ step {step + 1}: " + " ".join(f"{k}={v:.5f}" for k, v in metrics.items()), flush=True)
            if not (ckpt_path and is_rank0 and save_top_k and monitor and monitor in metrics):
                return
            val = metrics[monitor]
            if len(best) < save_top_k or val < best[-1][0]:
                epoch = step // max(1, int(getattr(self.model, "steps_per_epoch", 1) or 1))
                path = os.path.join(os.path.dirname(os.path.abspath(ckpt_path)), f"epoch={epoch}-{monitor}={val:.4f}.ckpt")
                self.save_checkpoint(path)
                best.append((val, path))
                best.sort(key=lambda t: t[0])
                for _, old in best[save_top_k:]:
                    if os.path.exists(old) and old != path:
                        os.remove(old)
                del best[save_top_k:]

        for step in range(int(self.model.global_step), self.max_steps):
            losses = self.train_step(batch_fn(step), step)
            if self.log_every and step % self.log_every == 0:
                print(f"step {step}: " + " ".join(f"{float(l):.5f}" for l in losses), flush=True)
            if ckpt_path and save_every and is_rank0 and (step + 1) % save_every == 0 and step + 1 < self.max_steps:
                self.save_checkpoint(ckpt_path)
            if val_fn is not None and val_every and ((step + 1) % val_every == 0 or step + 1 == self.max_steps):
                run_validation(step)
        if ckpt_path and is_rank0:
            self.save_checkpoint(ckpt_path)
