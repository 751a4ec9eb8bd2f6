#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs 4 and 5) on ONE MI355X -- the headline metric stays in bench.py.

    python bench_extra.py --workload triple   [--bs 32] [--steps K] [--warmup W]      # dqvae-triple-r-03-03 complete step
    python bench_extra.py --workload stage2   [--bs 32]                               # DQ-Transformer p6c18 train step
    python bench_extra.py --workload sampling [--bs 8]                                # AR sampling token-steps/s (K/V caches + graphs; prefix recompute)

Each prints one JSON line {"workload", "metric", "value", "unit", "ms_per_step", ...}.  Synthetic 256x256 half-flat images,
random-init weights of the shipped YAML architectures, bf16 compute / fp32 master weights, inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
if "sampling" in sys.argv:
    # HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (ROCm 7.2 default: 4, of which the sampler's lanes saw TWO:
    # kernels of streams that share a queue serialise).  Eight queues let four sampling lanes run four token steps at once
    # (profiles/r06_sampler_lanes.txt).  Sampling only: the triple-grain TRAINING step loses 15 % with eight queues (233.7 / 241.4 vs
    # 203.7 / 198.3 img/s, same box), the dual-grain and stage-2 steps are neutral.  Read when the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402


def _stage2_cpu_worker(threads, lc, lf):
    """child process: the oracle's StackGPT teacher-forced step (forward + losses + backward, torch CPU) at bs = 1 on `threads`
    host threads; random-init weights of the p6c18 architecture, random tokens of the benchmark's sequence lengths"""
    torch.set_num_threads(threads)
    from dynamicvectorquantization_amd import config as cfg
    from oracle import stackgpt as osg
    c = cfg.load_yaml(os.path.join(REPO, "configs/stage2/uncond_imagenet_p6c18.yml"))
    tp = c.model.params.transformer_config
    gpt = cfg.instantiate_from_config(tp)                       # parameters only (CPU tensors); the compute below is the oracle's
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in gpt.state_dict().items() if not k.endswith(".mask")}
    n_head = int(tp.params.n_head)
    g = torch.Generator().manual_seed(0)
    ri = lambda hi, n: torch.randint(0, hi, (1, n), generator=g)
    cc, fc, cp, fp = ri(1024, lc), ri(1024, lf), ri(256, lc), ri(1024, lf)
    cs, fs = torch.zeros(1, lc, dtype=torch.long), torch.ones(1, lf, dtype=torch.long)
    content_target = torch.cat([cc, fc], dim=1)[:, 1:]
    t0 = time.time()
    n = 0
    while True:
        out = osg.forward(sd, n_head, cc, fc, cp, fp, cs, fs, content_target=content_target, coarse_position_target=cp[:, 1:],
                          fine_position_target=fp)
        (out["position_loss"] + out["content_loss"]).backward()
        for v in sd.values():
            v.grad = None
        n += 1
        if time.time() - t0 > 15.0 or n >= 3:
            break
    print(json.dumps({"n": n, "sec": time.time() - t0, "tokens": lc + lf - 1}), flush=True)


def stage2_cpu_baseline(lc, lf, timeout_s=240):
    import subprocess
    threads = max(1, min(16, os.cpu_count() or 1))
    cmd = [sys.executable, os.path.abspath(__file__), "--stage2-cpu-worker", str(threads), str(lc), str(lf)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": round(rec["n"] * rec["tokens"] / rec["sec"], 2), "unit": "tokens/sec", "cores": threads, "kind": "port",
                "sample": f"{rec['n']} teacher-forced step(s) (forward + losses + backward) of the torch-CPU oracle StackGPT p6c18 at bs=1, "
                          f"T={rec['tokens']}, {threads} threads of {os.cpu_count()} host cores, {rec['sec']:.1f} s"}
    except Exception as e:                                      # the GPU numbers must not depend on the host baseline
        return {"value": None, "unit": "tokens/sec", "cores": threads, "kind": "port", "sample": f"failed: {type(e).__name__}"}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--stage2-cpu-worker":
        return _stage2_cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", required=True, choices=["triple", "stage2", "sampling"])
    ap.add_argument("--bs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--codebook", type=int, default=None, help="override the codebook size (BASELINE config 4 quotes 8192)")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "an MI355X is required"
    dev = torch.device("cuda", 0)
    from dynamicvectorquantization_amd import _lib, config as cfg, runtime as rt, synth
    from dynamicvectorquantization_amd.trainer import Trainer
    _lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
    rt.set_compute_dtype("bf16")
    os.chdir(REPO)
    torch.manual_seed(0)

    PEAK_BF16, PEAK_HBM = 2.5e15, 8.0e12

    def roofline_of_step(trainer, batch, idx):
        """per-kernel-family HIP-event timing of ONE eagerly launched step (like bench.py): the family the step spends most
        time in, its algorithmic flop / byte rate against the MI355X peaks"""
        from dynamicvectorquantization_amd import kernels as K
        K.profile_count_start()
        trainer.train_step(batch, idx)
        n = K.profile_count_stop()
        K.profile_prepare(n + 16)
        prev_side = os.environ.get("DVQ_SIDE_WGRAD")
        os.environ["DVQ_SIDE_WGRAD"] = "0"      # one stream: per-kernel brackets must not hold a concurrent kernel's time
        K.profile_start()
        trainer.train_step(batch, idx + 1)
        prof = K.profile_stop()
        if prev_side is None:
            os.environ.pop("DVQ_SIDE_WGRAD", None)
        else:
            os.environ["DVQ_SIDE_WGRAD"] = prev_side
        if not prof:
            return None, {}
        fam = {k: dict(launches=v["launches"], ms_per_step=round(v["ms"], 3),
                       TFLOPs=round(v["flops"] / max(1e-9, v["ms"] * 1e-3) / 1e12, 2),
                       GBps=round(v["bytes"] / max(1e-9, v["ms"] * 1e-3) / 1e9, 1)) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k]["ms"])
        v = prof[dom]
        ach = v["flops"] / (v["ms"] * 1e-3)
        return {"kernel": dom, "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16, 4), "traffic": None, "launches": v["launches"],
                "avg_launch_ms": round(v["ms"] / max(1, v["launches"]), 4),
                "timed": "HIP events around every launch of this kernel family during one eagerly launched step after the timed steps"}, fam

    def timed_steps(trainer, batches, steps, warmup):
        for i in range(warmup):
            trainer.train_step(batches[i % len(batches)], i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            trainer.train_step(batches[i % len(batches)], warmup + i)
        timed_steps.host_enqueue_s = time.perf_counter() - t0      # until the last step was handed over (no sync inside the loop)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    if args.workload == "triple":
        bs = args.bs or 32
        c = cfg.load_yaml(os.path.join(REPO, "configs/stage1/dqvae-triple-r-03-03_imagenet.yml"))
        if args.codebook:
            c.model.params.vqconfig.params.codebook_size = args.codebook
        model = cfg.instantiate_from_config(c.model).to(dev)
        model.learning_rate, model.training_steps, model.steps_per_epoch = 4.5e-6 * bs, 100000, 1000
        model.train()
        tr = Trainer(model, max_steps=args.steps)
        batches = [{"image": torch.from_numpy(synth.half_flat_images(bs, 256, seed=77 + i)).to(dev)} for i in range(2)]
        dt = timed_steps(tr, batches, args.steps, args.warmup)
        ind = model._last["grain"]
        hist = torch.bincount(ind.reshape(-1), minlength=3).tolist()
        roof, fam = roofline_of_step(tr, batches[0], args.warmup + args.steps)
        out = {"workload": "triple", "metric": "images/sec (256x256) DQ-VAE triple-grain train step, complete two-optimizer objective",
               "value": round(bs * args.steps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / args.steps * 1e3, 2),
               "config": {"yaml": "configs/stage1/dqvae-triple-r-03-03_imagenet.yml", "bs": bs,
                          "codebook": int(c.model.params.vqconfig.params.codebook_size),
                          "grain_histogram": hist, "step_graph_replays": tr.graph_replays},
               "roofline": roof, "kernel_families": fam,
               "steps": args.steps, "warmup": args.warmup, "dtype": "bf16", "data": "synthetic"}
    else:
        c = cfg.load_yaml(os.path.join(REPO, "configs/stage2/uncond_imagenet_p6c18.yml"))
        model = cfg.instantiate_from_config(c.model).to(dev)
        model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 5e-4, 0.0, 100000, 1000
        if args.workload == "stage2":
            bs = args.bs or 32
            model.train()
            tr = Trainer(model, max_steps=args.steps)
            batches = [{"image": torch.from_numpy(synth.half_flat_images(bs, 256, seed=177 + i)).to(dev)} for i in range(2)]
            dt = timed_steps(tr, batches, args.steps, args.warmup)
            with torch.no_grad():
                _, z = model.encode_to_z(batches[0]["image"])
            t_len = z["coarse_content"].shape[1] + z["fine_content"].shape[1] + 1
            n_par = sum(p.numel() for p in model.transformer.parameters())
            flops = 6.0 * n_par * bs * t_len + 12.0 * 24 * bs * t_len * t_len * 1024      # weights + attention (full square)
            roof, fam = roofline_of_step(tr, batches[0], args.warmup + args.steps)
            out = {"workload": "stage2", "metric": "images/sec DQ-Transformer (StackGPT p6c18) train step over frozen DQ-VAE codes",
                   "value": round(bs * args.steps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / args.steps * 1e3, 2),
                   "tokens_per_sec": round(bs * t_len * args.steps / dt, 1),
                   "host_enqueue_ms_per_step": round(getattr(timed_steps, "host_enqueue_s", 0.0) / args.steps * 1e3, 2),
                   "config": {"yaml": "configs/stage2/uncond_imagenet_p6c18.yml", "bs": bs, "seq_len": int(t_len),
                              "transformer_params": n_par, "dropout": 0.1},
                   "mfma_frac_est": round(flops * args.steps / dt / 2.5e15, 4), "roofline": roof, "kernel_families": fam,
                   "steps": args.steps, "warmup": args.warmup, "dtype": "bf16", "data": "synthetic"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = stage2_cpu_baseline(int(z["coarse_content"].shape[1]), int(z["fine_content"].shape[1]))
        else:
            model.eval()
            tr_ = model.transformer
            n_par = sum(p.numel() for p in tr_.parameters())
            n_layer_all = len(tr_.position_transformer) + len(tr_.content_transformer)
            n_embd = int(tr_.config.n_embd)

            def kv_run(bs):
                """end-to-end constrained sampling with K/V caches (fixed fine positions), random-init weights -> dict"""
                x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=277)).to(dev)
                with torch.no_grad():
                    cnd = model.encode_to_c(x)
                    model.sample_from_scratch(*cnd, sample=True, top_k=300, top_k_pos=100, process=False, fix_fine_position=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = model.sample_from_scratch(*cnd, sample=True, top_k=300, top_k_pos=100, process=False, fix_fine_position=True)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                ntok = int(r[0].shape[1] + r[1].shape[1])
                # HBM roofline of one token step: every transformer weight (bf16) is streamed once for the whole batch, plus the K and V
                # rows of the prefix (average prefix = half the sequence) of every block for every sequence
                w_bytes = 2.0 * n_par
                kv_bytes = bs * n_layer_all * 2 * (ntok / 2.0) * n_embd * 2.0
                steps_per_s = ntok / dt
                achieved = (w_bytes + kv_bytes) * steps_per_s
                return {"bs": bs, "tokens_per_sequence": ntok, "seconds": round(dt, 3), "token_steps_per_sec": round(bs * ntok / dt, 1),
                        "ms_per_token_step": round(dt / ntok * 1e3, 3),
                        "roofline": {"bound": "hbm", "weight_bytes_per_step": int(w_bytes), "kv_bytes_per_step_avg": int(kv_bytes),
                                     "achieved": round(achieved / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8e12, 4)}}

            def kv_run_lanes(bs, lanes=4, nb=8):
                """the same with `lanes` batches in flight (Dualformer.sample_many: one stream, K/V caches and captured token-step graphs
                per lane): nb independent batches of bs sequences, whole-job token-steps/s"""
                x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=277)).to(dev)
                kw = dict(sample=True, top_k=300, top_k_pos=100, process=False, fix_fine_position=True)
                with torch.no_grad():
                    conds = [model.encode_to_c(x) for _ in range(nb)]
                    model.sample_many(conds[:lanes], n_streams=lanes, **kw)           # captures every lane's graphs (untimed)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rs = model.sample_many(conds, n_streams=lanes, **kw)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                ntok = sum(int(r[0].shape[1] + r[1].shape[1]) for r in rs)
                nt1 = ntok / nb
                w_bytes = 2.0 * n_par
                kv_bytes = bs * n_layer_all * 2 * (nt1 / 2.0) * n_embd * 2.0
                # one "token step" of a lane still streams every weight once: lanes x (weights + K/V rows) per wall-clock step pair
                achieved = (w_bytes + kv_bytes) * (ntok / dt)
                return {"bs": bs, "lanes": lanes, "batches": nb, "seconds": round(dt, 3), "token_steps_per_sec": round(bs * ntok / dt, 1),
                        "ms_per_token_step_per_lane": round(dt / ntok * lanes * 1e3, 3),
                        "roofline": {"bound": "hbm", "weight_bytes_per_step": int(w_bytes), "kv_bytes_per_step_avg": int(kv_bytes),
                                     "achieved": round(achieved / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8e12, 4)}}

            sizes = [args.bs] if args.bs else [8, 50]             # 50 = the reference sampler's default (scripts/sample_val/sample_dynamic_uncond.py:29)
            res = {}
            bs = sizes[0]
            x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=277)).to(dev)
            with torch.no_grad():
                _, z = model.encode_to_z(x)
                tf = model.teacher_forcing_inputs(z, model.encode_to_c(x))
                cc, fc, cp, fp, cs, fs = (tf[k] for k in ("coarse_content", "fine_content", "coarse_position", "fine_position",
                                                         "coarse_seg", "fine_seg"))
                # one sampling step of the fine stream at (a) half and (b) the full prefix: position pass + content pass,
                # recomputed over the whole prefix exactly like the reference's sampler (no KV cache)
                for tag, lf in (("half_prefix", fc.shape[1] // 2), ("full_prefix", fc.shape[1] - 1)):
                    def step():
                        h, pl = tr_.sample_fine_position(cc, fc[:, :lf], cp, fp[:, :lf], cs, fs[:, :lf])
                        tr_.sample_fine_content(cc, fc[:, :lf], cp, fp[:, :lf + 1], cs, fs[:, :lf], position_hidden=h)
                    for _ in range(2):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 5
                    for _ in range(n):
                        step()
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / n
                    res[tag] = {"bs": bs, "prefix_len": int(cc.shape[1] + lf), "ms_per_token_step": round(dt * 1e3, 2),
                                "tokens_per_sec": round(bs / dt, 1)}
            runs = [kv_run(b_) for b_ in sizes]
            lane_runs = []
            if os.environ.get("DVQ_BENCH_LANES", "4") != "0":
                for b_ in sizes:
                    try:
                        lane_runs.append(kv_run_lanes(b_, int(os.environ.get("DVQ_BENCH_LANES", "4"))))
                    except Exception as e:          # secondary measurement: never costs the single-lane figures
                        lane_runs.append({"bs": b_, "failed": f"{type(e).__name__}: {str(e)[:160]}"})
            res["kv_cached_end_to_end"] = runs[0]
            for r_ in runs[1:]:
                res[f"kv_cached_end_to_end_bs{r_['bs']}"] = r_
            out = {"workload": "sampling", "metric": "AR sampling token-steps/sec (one position + one content token per step): end-to-end "
                                                     "with K/V caches; `*_prefix` = the reference's schedule (whole prefix recomputed)",
                   "value": runs[0]["token_steps_per_sec"], "unit": "token-steps/sec", "detail": res,
                   "by_batch": {str(r_["bs"]): {"token_steps_per_sec": r_["token_steps_per_sec"], "roofline": r_["roofline"]} for r_ in runs},
                   "by_batch_concurrent_lanes": {f"{r_['bs']}x{r_.get('lanes', 2)}": r_ for r_ in lane_runs},
                   "roofline": runs[0]["roofline"],
                   "config": {"yaml": "configs/stage2/uncond_imagenet_p6c18.yml", "bs": bs, "batch_sizes": sizes,
                              "transformer_params": n_par}, "dtype": "bf16", "data": "synthetic"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
