"""Which kernel gives run-to-run different results when two processes share the GPU?

Two processes run the stage-2 forward (frozen first stage -> codes -> StackGPT losses) and forward+backward on the same inputs several
times; every call into dynamicvectorquantization_amd.kernels is wrapped and the checksums of all tensors it touched are recorded after
a device synchronise.  The first call (in issue order) whose checksum moves between repeats is the suspect.
    python tools/debug/race_hunt.py [nproc=2] [reps=5]"""
import copy, inspect, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))


def worker(rank, nproc, reps):
    import torch
    torch.cuda.set_device(0)
    from dynamicvectorquantization_amd import kernels as K, runtime as rt, synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    import test_gpu_stage2 as S
    dev = torch.device("cuda:0")
    log = []

    def tensors(o, acc):
        if isinstance(o, torch.Tensor):
            acc.append(o)
        elif isinstance(o, (tuple, list)):
            for e in o:
                tensors(e, acc)
        elif isinstance(o, dict):
            for e in o.values():
                tensors(e, acc)
        return acc

    def checksum(t):
        if t.numel() == 0 or not t.is_cuda:
            return (0.0, 0.0)
        if t.is_floating_point():
            d = t.detach().double()
            return (float(d.sum()), float(d.abs().sum()))
        d = t.detach().to(torch.int64)
        return (float(d.sum()), float(d.abs().sum()))

    def wrap(name, fn):
        def w(*a, **kw):
            r = fn(*a, **kw)
            ts = tensors([r, a, kw], [])
            torch.cuda.synchronize()
            log.append((name, [tuple(t.shape) for t in ts], [checksum(t) for t in ts]))
            return r
        return w
    skip = {"lib", "arena_reset", "version"}
    for n, f in list(vars(K).items()):
        if inspect.isfunction(f) and f.__module__ == K.__name__ and not n.startswith("_") and n not in skip:
            setattr(K, n, wrap(n, f))

    cfg = copy.deepcopy(S.dualformer_config())
    cfg["params"]["transformer_config"]["params"].update(embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0)
    x = torch.from_numpy(synth.half_flat_images(32, 64, seed=511)).to(dev)
    with rt.compute_dtype_ctx(torch.float32):
        torch.manual_seed(0)
        model = instantiate_from_config(cfg).to(dev)
        model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 0.0, 100, 10
        model.train()
        tr = Trainer(model, max_steps=6)
        for phase in ("forward", "forward+backward"):
            runs = []
            for rep in range(-1, reps):          # rep -1: warm-up (weight packings, workspaces), not compared
                del log[:]
                K.arena_reset(dev)
                if phase == "forward":
                    with torch.no_grad():
                        o = model.shared_step({"image": x}, 0)
                else:
                    tr.buckets[0].zero()
                    with torch.autograd.set_multithreading_enabled(False), rt.side_wgrad():
                        loss = model.training_step({"image": x}, 0)
                        loss.backward()
                torch.cuda.synchronize()
                if rep >= 0:
                    runs.append(list(log))
            base = runs[0]
            print(f"[{rank}] {phase}: {len(base)} kernel-layer calls per repeat; lengths {[len(r) for r in runs]}", flush=True)
            seen = 0
            for rep in range(1, reps):
                first = None
                nbad = 0
                for i, (a, b) in enumerate(zip(base, runs[rep])):
                    bad = a[0] != b[0]
                    worst = 0.0
                    for ca, cb in zip(a[2], b[2]):
                        scale = max(abs(ca[1]), abs(cb[1]), 1e-30)
                        worst = max(worst, abs(ca[0] - cb[0]) / scale, abs(ca[1] - cb[1]) / scale)
                    if bad or worst > 3e-6:
                        nbad += 1
                        if first is None:
                            first = (i, a[0], a[1], worst)
                print(f"[{rank}] {phase} rep {rep}: {nbad} calls differ from rep 0; first: {first}", flush=True)
                if first is not None and seen < 2:
                    seen += 1
                    i = first[0]
                    for j in range(max(0, i - 2), min(len(base), i + 4)):
                        a, b = base[j], runs[rep][j]
                        moved = [k for k, (ca, cb) in enumerate(zip(a[2], b[2])) if abs(ca[1] - cb[1]) > 3e-6 * max(abs(ca[1]), abs(cb[1]), 1e-30)]
                        print(f"[{rank}]     #{j} {a[0]} {a[1]}  moved tensors {moved}  rep0 {[f'{c[1]:.9g}' for c in a[2]]}  rep{rep} {[f'{c[1]:.9g}' for c in b[2]]}", flush=True)


if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    if nproc == 1:
        worker(0, 1, reps)
    else:
        import torch.multiprocessing as mp
        mp.spawn(worker, args=(nproc, reps), nprocs=nproc, join=True)
