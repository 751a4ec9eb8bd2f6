"""World-size-2 gloo runs of the data-parallel CONTROL FLOW (no GPU, no kernels): the Trainer's two-optimizer step on a mock
module whose backward writes rank-dependent flat gradients and reports finished parameter sets through `_grad_hook`, the
initial parameter / buffer broadcast, the replica-equality check, the reference's learning-rate rule, and bench.py's own
rank spawning (`--dry-run`)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mock_classes():
    import torch.nn as nn
    from dynamicvectorquantization_amd.trainer import HipAdam

    class CpuAdam(HipAdam):
        """HipAdam's bookkeeping (flat buffers, groups, step counter) with the update done by torch on the CPU"""

        def prepare_step(self):
            self._prepared = True

        @torch.no_grad()
        def step(self, closure=None):
            st = self._fstate
            st["step"] += 1
            self._prepared = False
            for g, (off, n) in zip(self.param_groups, self._segments):
                sl = slice(off, off + n)
                b1, b2 = g["betas"]
                gr = self.flat.flat_g[sl]
                st["m"][sl].mul_(b1).add_(gr, alpha=1 - b1)
                st["v"][sl].mul_(b2).addcmul_(gr, gr, value=1 - b2)
                mh, vh = st["m"][sl] / (1 - b1 ** st["step"]), st["v"][sl] / (1 - b2 ** st["step"])
                self.flat.flat_p[sl].sub_(g["lr"] * mh / (vh.sqrt() + g["eps"]))

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, mod, oi, x, *params):
            ctx.mod, ctx.oi, ctx.n, ctx.shape = mod, oi, len(params), x.shape
            return x.sum() * 0 + 1.0

        @staticmethod
        def backward(ctx, g):
            m, rank = ctx.mod, dist.get_rank()
            if ctx.oi == 0:
                # "decoder side" first, then the encoder levels from the last to the first -- each announced through the hook
                for blk in (m.decoder, m.encoder[2], m.encoder[1], m.encoder[0]):
                    for i, p in enumerate(blk.parameters()):
                        p.grad.add_(float(rank + 1) * (i + 1))
                    if m._grad_hook is not None:
                        m._grad_hook(list(blk.parameters()))
                    m.hook_calls += 1
            else:
                for i, p in enumerate(m.disc.parameters()):
                    p.grad.add_(float(rank + 1) * 10 * (i + 1))
            return (None, None, torch.zeros(ctx.shape)) + (None,) * ctx.n

    class Mock(nn.Module):
        GRAPH_SAFE = False

        def __init__(self):
            super().__init__()
            mk = lambda n: nn.Sequential(nn.Linear(n, n), nn.Linear(n, 7))      # noqa: E731
            self.encoder = nn.ModuleList([mk(40), mk(300), mk(24)])
            self.decoder = mk(512)
            self.disc = mk(64)
            self.register_buffer("ema", torch.zeros(5))
            self.learning_rate, self.global_step, self.current_epoch = 1e-2, 0, 0
            self._grad_hook, self.hook_calls = None, 0

        def configure_optimizers(self):
            oa = CpuAdam(list(self.encoder.parameters()) + list(self.decoder.parameters()), lr=self.learning_rate, betas=(0.5, 0.9))
            od = CpuAdam(list(self.disc.parameters()), lr=self.learning_rate, betas=(0.5, 0.9))
            fn = lambda step: 1.0 / (1 + step)           # noqa: E731
            return [oa, od], [{"scheduler": torch.optim.lr_scheduler.LambdaLR(o, fn)} for o in (oa, od)]

        def training_step(self, batch, batch_idx, optimizer_idx):
            params = list(self.parameters())
            return _Fn.apply(self, optimizer_idx, batch["image"].requires_grad_(True), *params)

    return Mock


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DVQ_DP_CHECK_EVERY="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamicvectorquantization_amd import trainer as T
    ok, why = True, ""
    try:
        torch.manual_seed(100 + rank)                 # DIFFERENT initial weights per rank: the broadcast has to fix that
        model = _mock_classes()()
        model.ema.fill_(float(rank))
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **k):                          # every all-reduce on a gradient buffer: [begin, end) in its flat buffer
            for gb in getattr(spy, "buckets", []):
                base, n = gb.fp.flat_g.data_ptr(), gb.fp.flat_g.numel()
                if base <= t.data_ptr() < base + 4 * n:
                    calls.append((id(gb), (t.data_ptr() - base) // 4, (t.data_ptr() - base) // 4 + t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = spy
        tr = T.Trainer(model, max_steps=3, use_graph=False)
        spy.buckets = tr.buckets
        for gb in tr.buckets:
            gb.bucket_elems = 50000                   # several buckets per range
        w0 = [p.detach().clone() for p in model.parameters()]
        # initial broadcast: rank 0's weights and buffers everywhere
        gathered = [torch.zeros_like(tr.buckets[0].fp.flat_p) for _ in range(world)]
        dist.all_gather(gathered, tr.buckets[0].fp.flat_p)
        assert torch.equal(gathered[0], gathered[1]) and float(model.ema[0]) == 0.0, "initial broadcast"
        assert T.replicas_equal(model)
        lr0 = [g["lr"] for o in tr.opts for g in o.param_groups]
        tr.train_step({"image": torch.ones(2, 3)}, 0)
        # (1) every element of both gradient buffers was all-reduced exactly once in the step
        for gb in tr.buckets:
            spans = sorted((lo, hi) for g_, lo, hi in calls if g_ == id(gb))
            pos = 0
            for lo, hi in spans:
                assert lo == pos, ("gap or overlap", spans)
                pos = hi
            assert pos == gb.fp.flat_g.numel(), spans
        # (2) the hook fired from inside the backward: decoder + encoder levels were launched before the closing reduce()
        assert model.hook_calls == 4
        first_ae = [c for c in calls if c[0] == id(tr.buckets[0])][0]
        dec_lo, _ = tr.buckets[0].param_range(list(model.decoder.parameters()))
        assert first_ae[1] == dec_lo, (first_ae, dec_lo)
        # (3) gradients were AVERAGED over the ranks: (1 + 2) / 2 * (i + 1)
        for blk in (model.decoder, model.encoder[1]):
            for i, p in enumerate(blk.parameters()):
                assert torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))), "average"
        for i, p in enumerate(model.disc.parameters()):
            assert torch.allclose(p.grad, torch.full_like(p, 15.0 * (i + 1)))
        # (4) identical update on both ranks, LambdaLR stepped once per optimizer, replicas still equal
        assert all(not torch.equal(a, p.detach()) for a, p in zip(w0, model.parameters()))
        assert [g["lr"] for o in tr.opts for g in o.param_groups] == [l * 0.5 for l in lr0]
        assert T.replicas_equal(model) and model.global_step == 1
        tr.train_step({"image": torch.ones(2, 3)}, 1)
        # (5) a diverged replica is detected by the periodic check
        if rank == 1:
            with torch.no_grad():
                next(model.disc.parameters()).add_(1e-3)
        try:
            tr.train_step({"image": torch.ones(2, 3)}, 2)
            raise AssertionError("divergence not detected")
        except RuntimeError as e:
            assert "diverged" in str(e)
        # (6) the reference's learning-rate rule (train.py:248-257): ngpu * batch_size * base_lr
        assert T.reference_learning_rate({"base_learning_rate": 4.5e-6}, world, 64) == 2 * 64 * 4.5e-6
        assert T.reference_learning_rate({"learning_rate": 5e-4}, world, 30) == 5e-4
    except Exception as e:      # noqa: BLE001
        import traceback
        ok, why = False, traceback.format_exc()
    q.put((rank, ok, why))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_optimizer_step_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), "\n".join(w for _, _, w in res)


def test_bench_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` (no launcher): bench.py starts the two ranks itself, they rendezvous, run the timed loop with
    the barrier / max-over-ranks contract and rank 0 prints ONE JSON line (CPU dry run: gloo, no kernels)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True
    assert rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 128 and rec["value"] > 0
    assert rec["config"]["learning_rate"] == pytest.approx(2 * 64 * 4.5e-6)
