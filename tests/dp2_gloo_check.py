"""Run by tests/test_gpu_dp2.py in child processes: the data-parallel training step with TWO ranks and the real kernels.

RCCL refuses two ranks on one device ("Duplicate GPU detected") and the builder's boxes have one GPU, so the two ranks share
cuda:0 and exchange device tensors through the gloo backend (it stages them through the host): every piece of the step that the
one-rank nccl test leaves a no-op is live here -- gradients pre-divided by 2 and summed over ranks from inside the backward, the
side-stream join before a range is exchanged, the VQ-EMA statistics all-reduce, graph breaks around the collectives and launch-list
replays between them, the initial broadcast from rank 0.

`python tests/dp2_gloo_check.py dp PORT OUT`   spawns ranks 0 and 1, each training on its half of every batch
`python tests/dp2_gloo_check.py dp_full PORT OUT`  the same with the complete objective (both optimizers, LPIPS, PatchGAN)
`python tests/dp2_gloo_check.py single 0 OUT`  one process, no process group, the whole batch
`python tests/dp2_gloo_check.py dp_s2 | single_s2 ...`  the same pair for the stage-2 (StackGPT) step
`python tests/dp2_gloo_check.py vq_share 0 -`  two processes + a copy stream share the GPU while the code search must stay exact
Both write {losses, parameters, Adam moments, EMA buffers} of the autoencoder-only objective (mean losses: the average of the two
half-batch gradients IS the full-batch gradient) to OUT (rank 0 only); the parent compares."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
STEPS = 7
# DVQ_DP2_BACKEND=nccl: one device per rank and the real RCCL backend (boxes with >= 2 GPUs: tests/test_gpu_dp2.py picks it when it
# sees them); default gloo on one shared device
BACKEND = os.environ.get("DVQ_DP2_BACKEND", "gloo")


def say(*a):
    print(*a, flush=True)


def run(rank, world, port, out, loss="ae"):
    import numpy as np
    import torch
    import torch.distributed as dist
    di = rank if (BACKEND == "nccl" and world > 1) else 0
    torch.cuda.set_device(di)
    if world > 1:
        dist.init_process_group(BACKEND, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        if BACKEND == "nccl":                 # the group works at all (the parent skips -- not fails -- when RCCL cannot come up on this box)
            probe = torch.ones(4, device=f"cuda:{di}")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert float(probe[0]) == world
            say(f"DP2_PG_OK {rank}")
    from dynamicvectorquantization_amd import runtime as rt, synth
    import test_gpu_stepgraph as T
    dev = torch.device("cuda", di)
    full = [torch.from_numpy(synth.half_flat_images(4, 64, seed=140 + i)).to(dev) for i in range(3)]
    with rt.compute_dtype_ctx(torch.float32):
        # different seeds per rank: the Trainer's start-up broadcast (not equal seeding) must make the replicas identical
        model, tr = T._make(dev, True, loss, seed=0 if world == 1 else 11 * rank)
        # dead-code restarts draw rows of the LOCAL batch: a data-parallel run and a single-process run restart with other rows by
        # construction (the reference too) -- off for the comparison, exercised by test_vq_ema_golden / the gloo CPU tests
        model.quantize.codebook.restart_unused_codes = False
        losses = []
        for i in range(STEPS):
            x = full[i % 3]
            if world > 1:
                x = x[2 * rank: 2 * rank + 2].contiguous()
            o = tr.train_step({"image": x}, i)
            losses.append([float(l) for l in o])
        torch.cuda.synchronize()
    say(f"rank {rank} of {world}: replays {tr.graph_replays}, segments {tr._graph['sg'].n_segments() if tr._graph else 0}, "
        f"launched {[b.launched for b in tr.buckets]}, last loss {losses[-1]}")
    if world > 1:
        assert tr.graph_replays == STEPS - 2 and tr._graph is not None
        kinds = [k for k, _ in tr._graph["sg"].items]
        assert kinds.count("eager") >= 3 and kinds.count("list") == kinds.count("eager") + 1, kinds    # VQ exchange, gradient ranges, wait
        assert all(b.launched > 0 for b in (tr.buckets if loss == "full" else tr.buckets[:1]))
        # the replicas stayed identical: checksums of all parameters and EMA buffers agree across ranks
        sums = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                           [model.state_dict()[k].double().sum() for k in ("quantize.codebook.cluster_size_ema", "quantize.codebook.embed_ema")])
        both = [torch.zeros_like(sums) for _ in range(2)]
        dist.all_gather(both, sums)
        assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())
    if rank == 0:
        sd = {"losses": np.array(losses)}
        for n, p in model.named_parameters():
            sd["p:" + n] = p.detach().float().cpu().numpy()
        for k in ("quantize.codebook.cluster_size_ema", "quantize.codebook.embed_ema"):
            sd["b:" + k] = model.state_dict()[k].float().cpu().numpy()
        sd["adam_m"] = tr.opts[0]._fstate["m"].float().cpu().numpy()
        np.savez(out, **sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    say(f"DP2_RANK_OK {rank}")


def run_s2(rank, world, port, out):
    """stage 2 (Dualformer / StackGPT, eager steps): per-block gradient ranges exchanged from inside the backward while the Linear
    weight gradients of that block run on the side stream -- the join before a range is pre-divided and summed is what this checks.
    Both ranks train on the SAME 32 images and the single process on those 32 twice: equal token counts per rank, so the mean of
    the two rank losses is the whole-batch loss.  Dropout off, fp32."""
    import copy
    import numpy as np
    import torch
    import torch.distributed as dist
    di = rank if (BACKEND == "nccl" and world > 1) else 0
    torch.cuda.set_device(di)
    if world > 1:
        dist.init_process_group(BACKEND, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from dynamicvectorquantization_amd import runtime as rt, synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    import test_gpu_stage2 as S
    dev = torch.device("cuda", di)
    cfg = copy.deepcopy(S.dualformer_config())
    cfg["params"]["transformer_config"]["params"].update(embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0)
    half = torch.from_numpy(synth.half_flat_images(32, 64, seed=511)).to(dev)
    x = half if world > 1 else torch.cat([half, half])
    with rt.compute_dtype_ctx(torch.float32):
        torch.manual_seed(0 if world == 1 else 7 * rank)
        model = instantiate_from_config(cfg).to(dev)
        if world > 1:
            # the frozen first stage is not a trained parameter set: give every rank the single process's (seed 0) weights
            torch.manual_seed(0)
            ref = instantiate_from_config(cfg)
            model.first_stage_model.load_state_dict(ref.first_stage_model.state_dict())
            if rank == 0:
                model.transformer.load_state_dict(ref.transformer.state_dict())          # rank 0 = the reference initialisation
            del ref
        model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 0.0, 100, 10
        model.train()
        tr = Trainer(model, max_steps=6)
        if os.environ.get("DVQ_DP2_DIAG"):              # forward-only repeats: is a difference there before any training, and does it move within a process?
            for rep in range(3):
                with torch.no_grad():
                    _, z0 = model.encode_to_z(x)
                    o0 = model.shared_step({"image": x}, 0)
                say(f"diag rank {rank}/{world} rep {rep}: codes {[int(z0[k].double().sum()) for k in sorted(z0) if z0[k] is not None]} "
                    f"content {float(o0['content_loss']):.8f} position {float(o0['position_loss']):.8f} "
                    f"psum {float(sum(p.double().abs().sum() for p in model.transformer.parameters())):.6f}")
        losses = [float(tr.train_step({"image": x}, i)[0]) for i in range(5)]
        torch.cuda.synchronize()
    with torch.no_grad():
        _, z = model.encode_to_z(half)
    ntok = int(half.shape[0] * (z["coarse_content"].shape[1] + z["fine_content"].shape[1] + 1))
    say(f"s2 rank {rank} of {world}: tokens per rank batch {ntok}, launched {[b.launched for b in tr.buckets]}, losses {losses}")
    assert ntok >= 1024, ntok                                   # the side-stream weight-gradient path needs >= 1024 rows
    if world > 1:
        assert tr.buckets[0].launched > 0
    if rank == 0:
        sd = {"losses": np.array(losses)}
        for n, p in model.transformer.named_parameters():
            sd["p:" + n] = p.detach().float().cpu().numpy()
        np.savez(out, **sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    say(f"DP2_RANK_OK {rank}")


def run_vq_share(rank, world, port, out):
    """the code search while the GPU is SHARED: a second process runs the same loop and a side stream of this one streams large
    copies, so DMA latencies inside the kernels are long and irregular.  Every result must still be the exact argmin.  (What this
    caught: the D = 64 search kernels passed a stage barrier with their own LDS-DMA pieces in flight -- correct on an idle GPU, a few
    near-tied rows wrong under load.  The structural guard is tools/lint_dma_barriers.py; this is the stress test beside it.)"""
    import numpy as np
    import torch
    torch.cuda.set_device(0)
    from dynamicvectorquantization_amd import kernels as K
    from oracle import vq as ovq
    dev = torch.device("cuda:0")
    hog_src = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    ncalls = 0
    for (n, d, k) in ((2048, 64, 512), (4096 + 96, 64, 1024), (2048, 128, 512), (4096, 256, 1024)):
        g = torch.Generator().manual_seed(1000 * n + d + rank)
        x = torch.randn(n, d, generator=g)
        cb = (torch.randn(k, d, generator=g) * 0.7)
        cb[7] = cb[3]                                                  # an exact tie: the lowest index must win
        xb = x.to(torch.bfloat16)
        want32 = torch.from_numpy(ovq.argmin_exact(x.numpy(), cb.numpy())).to(dev)
        want16 = torch.from_numpy(ovq.argmin_exact(xb.float().numpy(), cb.numpy())).to(dev)
        x, xb, cb = x.to(dev), xb.to(dev), cb.to(dev)
        prep = K.vq_prepare(cb)
        for rep in range(60):
            if rep % 4 == 0:
                with torch.cuda.stream(side):
                    hog_dst.copy_(hog_src)
            bad += (K.vq_argmin(x, cb, prep) != want32).sum()
            bad += (K.vq_argmin(xb, cb, prep) != want16).sum()
            ncalls += 2
    torch.cuda.synchronize()
    say(f"vq_share rank {rank}: {ncalls} searches, {int(bad)} wrong indices")
    assert int(bad) == 0, int(bad)
    say(f"DP2_RANK_OK {rank}")


def main():
    mode, port, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    if mode == "vq_share":
        import torch.multiprocessing as mp
        mp.spawn(run_vq_share, args=(2, port, out), nprocs=2, join=True)
        say("DP2_OK")
        return
    if mode == "single_s2":
        run_s2(0, 1, 0, out)
    elif mode == "dp_s2":
        import torch.multiprocessing as mp
        mp.spawn(run_s2, args=(2, port, out), nprocs=2, join=True)
    elif mode == "single":
        run(0, 1, 0, out)
    else:
        # "dp": autoencoder-only objective (compared with the single process); "dp_full": the complete two-optimizer objective
        # (LPIPS + PatchGAN; its BatchNorm statistics are per rank, so only the replicas are compared with each other)
        import torch.multiprocessing as mp
        mp.spawn(run, args=(2, port, out, "full" if mode == "dp_full" else "ae"), nprocs=2, join=True)
    say("DP2_OK")


if __name__ == "__main__":
    main()
