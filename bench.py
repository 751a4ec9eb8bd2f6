#!/usr/bin/env python3
"""Headline benchmark: DQ-VAE (dual-grain, entropy-routed) train-step images/sec on synthetic 256x256
batches + VQ-argmin GB/s, on N MI355X of one node (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: entropy gate -> encoder -> quant_conv -> VQ (argmin,
EMA codebook update) -> post_quant_conv -> decoder -> loss -> full backward -> (RCCL gradient all-reduce) ->
Adam, in bf16 activations / fp32 master weights.  The default objective is the shipped config's complete
two-optimizer step (L1 + LPIPS + adaptive hinge GAN + codebook on the autoencoder, then a second autoencoder
forward and the PatchGAN hinge step -- exactly the work Lightning schedules for the reference); `--objective ae`
times the autoencoder-only step (L1 + codebook) and says so in `config.objective`.
Weak scaling: bs/GPU is fixed, value = global images / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "images/sec (256x256) DQ-VAE train step + VQ argmin GB/s, 1/2/4/8 MI355X"
PEAK_BF16 = 2.5e15      # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12
AE_TRAIN_FLOP_PER_IMG = 1180.8e9   # 3 x 393.6 GFLOP (conv + attention + VQ), 256x256 dual config
# complete objective: + second AE forward (393.6) + LPIPS (VGG16 40.1 GFLOP/img: 2 forwards + 1 dgrad) + PatchGAN ndf=64
# (6.29 GFLOP/img forward: generator branch fwd+dgrad, discriminator branch 2 x (fwd+dgrad+wgrad))
STEP_FLOP_PER_IMG = {"ae": AE_TRAIN_FLOP_PER_IMG, "full": AE_TRAIN_FLOP_PER_IMG + 393.6e9 + 3 * 40.1e9 + 8 * 6.29e9}
THR_JSON = os.path.join(REPO, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json")


PEAK_F32 = 157.3e12     # dense MFMA fp32 (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md
YAML = "configs/stage1/dqvae-entropy-dual-r05_imagenet.yml"


def full_config(objective="full", bs=64):
    """the `model:` section of the shipped YAML through the drop-in boundary (config.load_yaml + dotlist merge = train.py:109-111);
    the only overrides are data.params.batch_size and, for `--objective ae`, the two loss weights that switch LPIPS / GAN off"""
    from dynamicvectorquantization_amd import config as cfg
    return cfg.stage1_config(YAML, batch_size=bs, objective=objective).model


def _cpu_baseline_worker(threads, bs, objective="full", budget_s=15.0):
    """child process: time the oracle on `threads` host threads for ~budget_s seconds; prints one JSON line.
    objective: "full" / "ae" = train steps of the DQ-VAE objective at 256x256; "vq" = the reference's VQ argmin formula
    (fp32 addmm distances + argmin, quantize2_mask.py:29-55) at the BASELINE shape N=65536, D=256, K=1024."""
    torch.set_num_threads(threads)
    from dynamicvectorquantization_amd import synth
    if objective == "vq":
        from oracle import vq as ovq
        x, cb = synth.vq_inputs(65536, 256, 1024, "normal", 0)
        ovq.argmin_fp32_formula(x[:4096], cb)               # thread pool / allocator warm-up
        t0 = time.time()
        n = 0
        while True:
            ovq.argmin_fp32_formula(x, cb)
            n += 1
            if time.time() - t0 > budget_s or n >= 20:
                break
        print(json.dumps({"n": n, "sec": time.time() - t0}), flush=True)
        return
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from oracle import entropy as oent
    from oracle import train_step as ots
    torch.manual_seed(0)
    model = instantiate_from_config(full_config(objective))          # reference-identical init + key names (CPU tensors)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    param_keys = [k for k, _ in model.named_parameters()]
    del model
    thr = oent.threshold_from_table(THR_JSON, 0.5)
    x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=1234))
    # the reference draws torch.randperm(N) per EMA update for the code restart (quantize2_mask.py:97); here: one seeded permutation
    perm = np.random.RandomState(0).permutation(bs * (256 // 8) ** 2)
    t0 = time.time()
    n = 0
    while True:
        # oracle.train_step is pinned against the reference's own two-optimizer step (tests/golden/train_step_*.npz): both
        # training-mode autoencoder forwards with their EMA codebook updates, LPIPS, adaptive GAN weight, both Adam updates
        if objective == "full":
            ots.full_objective_steps(state, param_keys, [x], thr, steps=1, restart_perm=perm)
        else:
            ots.ae_only_steps(state, param_keys, [x], thr, steps=1, restart_perm=perm)
        n += 1
        if time.time() - t0 > budget_s or n >= 10:      # ~budget_s of CPU work
            break
    print(json.dumps({"n": n, "sec": time.time() - t0}), flush=True)


def _cpu_leg(objective, threads, bs, budget_s, timeout_s):
    """one bounded child process of the CPU oracle -> {"value", "unit", "cores", "sample"}"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), str(bs), objective, str(budget_s)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="")
    what = {"full": f"train step(s) of the complete two-optimizer objective at bs={bs}, 256x256 fp32",
            "ae": f"autoencoder-only train step(s) (L1 + codebook: forward + backward + Adam) at bs={bs}, 256x256 fp32",
            "vq": "VQ argmin call(s) in the reference's fp32 addmm formula at N=65536, D=256, K=1024"}[objective]
    unit = "GB/s (algorithmic bytes: rows + codebook + indices)" if objective == "vq" else "images/sec"
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        if objective == "vq":
            val = rec["n"] * (65536 * 256 * 4 + 1024 * 256 * 4 + 65536 * 8) / rec["sec"] / 1e9
        else:
            val = rec["n"] * bs / rec["sec"]
        return {"workload": {"full": "complete_step", "ae": "ae_only_step", "vq": "vq_argmin"}[objective], "value": round(val, 4),
                "unit": unit, "cores": threads, "kind": "port",
                "sample": f"{rec['n']} {what}, torch-CPU / numpy oracle, {threads} thread(s) of {os.cpu_count()} host cores, {rec['sec']:.1f} s"}
    except Exception as e:      # timeout or failure: report, never stall the GPU measurement
        return {"workload": objective, "value": None, "unit": unit, "cores": threads, "kind": "port",
                "sample": f"cpu baseline leg did not finish within {timeout_s}s ({type(e).__name__})"}


def cpu_baseline(objective="full", timeout_s=150):
    """oracle (a port of the reference's CPU path) timed on this host's cores.  Headline leg: the benchmark's own objective at
    bs = 1 on at most 16 threads (the torch-CPU conv path stops scaling / collapses under oversubscription well before the box's
    full core count).  `legs`: VQ argmin at the BASELINE shape on 16 threads and on ONE thread, the autoencoder-only
    step at bs = 2 on 16 threads and at bs = 1 on one thread.  Every leg is a bounded child process (~8-15 s of CPU work)."""
    threads = max(1, min(16, os.cpu_count() or 1))
    t_end = time.time() + timeout_s
    head = _cpu_leg(objective, threads, 1, 15.0, max(20.0, min(60.0, t_end - time.time())))
    legs = []
    for obj, th, bs, budget in (("vq", threads, 0, 6.0), ("vq", 1, 0, 6.0), ("ae", threads, 2, 8.0), ("ae", 1, 1, 10.0)):
        left = t_end - time.time()
        if left < 20.0:
            legs.append({"workload": obj, "cores": th, "skipped": "time budget"})
            continue
        legs.append(_cpu_leg(obj, th, bs, budget, min(left, 60.0)))
    out = {k: head[k] for k in ("value", "unit", "cores", "kind", "sample")}
    out["legs"] = legs
    return out


OBJECTIVES = {
    "full": "complete reference step, both optimizers: (0) entropy gate + encoder + VQ(argmin+EMA) + decoder, loss = L1 + LPIPS(VGG16, "
            "random weights) + adaptive-weight hinge GAN (PatchGAN ndf=64) + codebook, full backward + Adam; (1) second autoencoder "
            "forward, PatchGAN hinge loss on (x, xrec), backward + Adam",
    "ae": "autoencoder-only step: entropy gate + encoder + VQ(argmin+EMA) + decoder, loss = L1 + codebook, full backward + Adam "
          "(LPIPS and PatchGAN terms switched off: perceptual_weight = disc_factor = 0)",
}


def _pmc_traffic(family):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/*_bench_pmc.json)"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_bench_pmc.json")))
    if not files:
        return None
    try:
        rec = json.load(open(files[-1]))
        ks = {k: v for k, v in rec["kernels"].items() if k == family or k.startswith(family + "<")}
        if not ks:
            return None
        n = sum(v["launches"] for v in ks.values())
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from pmc_summarise import csrc_sha16
        now = csrc_sha16()
        return {"hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks.values()) / n),
                "source": "NOT measured in this run (PMC passes need rocprofv3 around the process): profiles/" + os.path.basename(files[-1]) +
                          f", taken at commit {rec.get('commit') or '?'} with kernel sources {rec.get('csrc_sha16') or '?'}; this run's kernel "
                          f"sources {now} ({'same' if now == rec.get('csrc_sha16') else 'DIFFERENT'}).  " + rec.get("_method", "")}
    except Exception:
        return None


def vq_microbench(dev, reps=20, in_training=None):
    """VQ argmin at the BASELINE shape (N=65536, D=256, K=1024): algorithmic GB/s, inputs resident in HBM.  Cases: N(0,1) rows
    and codes ("normal", fp32 and bf16 rows), the untrained-encoder distribution (|z|^2 ~ 12, codebook U(+-1/1024): tiny score
    gaps, many fp64 re-ranks), and -- when given -- the rows and codebook the training run of this process just saw."""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import synth
    out = {}
    cases = [("f32", "normal", torch.float32), ("bf16", "normal", torch.bfloat16), ("encoder_bf16", "encoder", torch.bfloat16)]
    for tag, dist_, dtype in cases + ([("in_training_bf16", None, None)] if in_training is not None else []):
        if dist_ is None:
            xt, cbt = in_training
        else:
            x, cb = synth.vq_inputs(65536, 256, 1024, dist_, 0)
            xt = torch.from_numpy(x).to(dev).to(dtype)
            cbt = torch.from_numpy(cb).to(dev)
        n, d = xt.shape
        k = cbt.shape[0]
        prep = K.vq_prepare(cbt)
        for _ in range(3):
            K.vq_argmin(xt, cbt, prep, impl=2)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            idx, flagged = K.vq_argmin(xt, cbt, prep, impl=2, return_flagged=True)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        nbytes = n * d * xt.element_size() + k * d * 4 + n * 8
        passes = 3 if xt.dtype == torch.float32 else 2          # bf16 MFMA passes of the split product
        out[tag] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 2), "frac_hbm_peak": round(nbytes / (ms * 1e-3) / PEAK_HBM, 4),
                    "TFLOPs_equiv": round(2 * n * k * d / ms / 1e9, 2),
                    "mfma_frac": round(passes * 2 * n * k * d / (ms * 1e-3) / PEAK_BF16, 4), "mfma_passes": passes,
                    "rerank_rows_full": int(flagged[0].item()), "rerank_rows_candidates": int(flagged[1].item()),
                    "rerank_rows_wide": int(flagged[2].item()),
                    "alg_bytes": nbytes, "shape": [n, d, k]}
    return out


def parity_bf16(model, x):
    """bf16 is the benchmark's precision, fp32 the parity path ("bf16 is perf mode with a reported mismatch
    rate").  The SAME weights (the ones the timed steps left behind), one bs-64 batch, eval forward through both instantiations of
    the kernels: how many code indices / grain decisions differ and how far the reconstructions are apart
    (reference bar for the indices: quantize2_mask.py:50-55)."""
    from dynamicvectorquantization_amd import runtime as rt
    was_training = model.training
    model.eval()
    res = {}
    try:
        with torch.no_grad():
            for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
                with rt.compute_dtype_ctx(dt):
                    o = model.ae_fwd(x, None)
                    res[tag] = (o["codes"].clone(), o["grain"].clone(), o["rec"].float().clone(), float(o["qloss"]))
        torch.cuda.synchronize()
    finally:
        model.train(was_training)
    cb, gb, rb, qb = res["bf16"]
    cf, gf, rf, qf = res["fp32"]
    # codes live on the fine grid; a coarse cell holds one code four times -- count CELLS (a coarse cell once)
    rep = gf.repeat_interleave(cb.shape[1] // gf.shape[1], 1).repeat_interleave(cb.shape[2] // gf.shape[2], 2).bool()
    diff = cb != cf
    n_fine = int(rep.sum())
    n_coarse = int((~rep).sum()) // 4
    mism = int((diff & rep).sum()) + int((diff & ~rep).sum()) // 4
    return {"batch": int(x.shape[0]), "cells": n_fine + n_coarse, "code_mismatches": mism,
            "code_mismatch_rate": round(mism / max(1, n_fine + n_coarse), 6),
            "grain_mismatches": int((gb != gf).sum()), "grain_cells": int(gf.numel()),
            "recon_rel_err": round(float((rb - rf).norm() / rf.norm()), 6),
            "qloss_bf16": round(qb, 6), "qloss_fp32": round(qf, 6),
            "note": "eval forward (entropy gate, encoder, VQ argmin, decoder) of the bf16 benchmark path vs the fp32 instantiation "
                    "of the same kernels on the weights the timed steps produced; indices are the exact argmin in both, they "
                    "differ where bf16 rounding of the encoder activations moves a row across a Voronoi boundary"}


def parity_vs_reference(dev):
    """the benchmark's precision scored against the REFERENCE'S OWN outputs: BASELINE config 1 = the shipped
    YAML at full width on 64 x 64 images, bs 2, parameters regenerated from synth (tests/golden/dqvae_c1.npz holds what the reference
    model of /root/reference returned for them -- tools/gen_golden.py).  `spread`: a codebook with score gaps far above rounding (the
    reference's indices are reproducible); `refinit`: the reference's U(+-1/K) initialisation (every gap at fp32 rounding level).
    north_star bar: indices exact, reconstruction 1e-3 relative."""
    from dynamicvectorquantization_amd import config as cfg
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    g = np.load(os.path.join(REPO, "tests", "golden", "dqvae_c1.npz"), allow_pickle=False)
    keys = [str(k) for k in g["state_keys"]]
    shapes = [tuple(int(v) for v in str(t).split(",")) if str(t) else () for t in g["state_shapes"]]
    geom = synth.DQVAE_GEOM["c1"]
    x = torch.from_numpy(synth.half_flat_images(2, 64, seed=4321)).to(dev)
    out = {"fixture": "tests/golden/dqvae_c1.npz (reference DualGrainVQModel.forward, fp32 CPU)", "geometry": "config 1: ch 128, "
           "codebook 1024 x 256, 64 x 64 images, bs 2 (128 code cells)"}
    for tag, dt in (("bf16", "bf16"), ("fp32", "fp32"), ("fp32x3", "fp32x3")):
        res = {}
        for variant in ("spread", "refinit"):
            model = cfg.instantiate_from_config(cfg.stage1_config(YAML, objective="none", geometry=geom).model)
            sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.dqvae_golden_state(keys, shapes, variant, geom["k"], geom["zc"]).items()}
            model.load_state_dict(sd)
            model = model.to(dev).eval()
            with torch.no_grad(), rt.compute_dtype_ctx(dt):
                rec, qloss, grain, _, _ = model(x)
                codes = model._last["codes"].cpu().numpy().astype(np.int32).reshape(-1)
            ref_codes, ref_rec = g[f"{variant}_codes"].reshape(-1), g[f"{variant}_rec"]
            recn = rec.float().cpu().numpy()
            bad = np.nonzero(codes != ref_codes)[0]
            r = {"code_mismatches": int(len(bad)), "cells": int(codes.size),
                 "grain_mismatches": int((grain.cpu().numpy().astype(np.int8) != g[f"{variant}_grain"]).sum()),
                 "recon_rel_err": round(float(np.linalg.norm(recn - ref_rec) / np.linalg.norm(ref_rec)), 6),
                 "recon_max_abs_err": round(float(np.abs(recn - ref_rec).max()), 6),
                 "qloss_rel_err": round(abs(float(qloss) - float(g[f"{variant}_qloss"])) / max(1e-12, abs(float(g[f"{variant}_qloss"]))), 6)}
            if variant == "refinit" and len(bad):
                r["max_exact_top2_gap_of_mismatched"] = float(g[f"{variant}_gap"][bad].max())
            # the decoder alone: the REFERENCE'S codes through this precision's codebook lookup + post_quant_conv + decoder (a flipped
            # code replaces a whole latent cell, so recon_rel_err above jumps with every mismatch; this one has none by construction)
            with torch.no_grad(), rt.compute_dtype_ctx(dt):
                q = model.quantize.get_codebook_entry(torch.from_numpy(ref_codes.reshape(g[f"{variant}_codes"].shape).astype(np.int64)).to(dev))
                rec2 = model.decode(q.permute(0, 3, 1, 2).contiguous().float()).float().cpu().numpy()
            r["recon_rel_err_decoder_on_reference_codes"] = round(float(np.linalg.norm(rec2 - ref_rec) / np.linalg.norm(ref_rec)), 6)
            res[variant] = r
            del model
        out[tag] = res
    return out


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves (same env contract as
    torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), forward rank 0's JSON line"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    if rc != 0:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def _dry_run(args, world, rank):
    """the multi-rank launch contract without a GPU: rendezvous (gloo), W untimed + K timed no-op steps between barriers, max over
    ranks, one JSON line from rank 0 -- what the driver's `--gpus N` command line exercises before any kernel runs"""
    from dynamicvectorquantization_amd.trainer import reference_learning_rate
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.zeros(1 << 16)

    def step():
        buf.add_(1.0)
        if world > 1:
            dist.all_reduce(buf)
            buf.div_(world)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(world * args.bs * args.steps / float(dt), 2), "unit": "images/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(dt) / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "DRY RUN (no GPU work): launch / rendezvous / timing contract only",
                                     "global_batch": world * args.bs, "parallelism": f"dp{world}",
                                     "learning_rate": reference_learning_rate({"base_learning_rate": 4.5e-6}, world, args.bs)}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extra_workloads(budget_s):
    """compact {value, unit, ms_per_step, roofline} blocks of the secondary workloads (BASELINE configs 4 and 5): the triple-grain
    DQ-VAE step with the 8192-entry codebook, the DQ-Transformer p6c18 train step and K/V-cached sampling, each measured by
    bench_extra.py in a child process; a workload that does not fit the remaining budget is reported as skipped"""
    import subprocess
    t_end = time.time() + budget_s
    jobs = [("triple_k8192", ["--workload", "triple", "--codebook", "8192", "--steps", "4", "--warmup", "4"]),
            ("stage2_p6c18", ["--workload", "stage2", "--steps", "3", "--warmup", "2"]),
            ("sampling_p6c18", ["--workload", "sampling"])]
    res = {}
    for name, extra in jobs:
        left = t_end - time.time()
        if left < 25.0:
            res[name] = {"skipped": "time budget"}
            continue
        try:
            r = subprocess.run([sys.executable, os.path.join(REPO, "bench_extra.py"), "--no-cpu-baseline"] + extra, capture_output=True,
                               text=True, timeout=left)
            e = json.loads(r.stdout.strip().splitlines()[-1])
            roof = e.get("roofline") or {}
            res[name] = {"metric": e.get("metric"), "value": e.get("value"), "unit": e.get("unit"), "ms_per_step": e.get("ms_per_step"),
                         "config": e.get("config"), "mfma_frac_est": e.get("mfma_frac_est"),
                         "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")} if roof else None}
            if "detail" in e:
                res[name]["detail"] = e["detail"]
            if "by_batch" in e:
                res[name]["by_batch"] = e["by_batch"]
            if "by_batch_concurrent_lanes" in e:
                res[name]["by_batch_concurrent_lanes"] = e["by_batch_concurrent_lanes"]
        except Exception as ex:                     # the headline line must not depend on a secondary workload
            res[name] = {"failed": f"{type(ex).__name__}: {str(ex)[:120]}"}
    return res


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        return _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "full",
                                    float(sys.argv[5]) if len(sys.argv) > 5 else 15.0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bs", type=int, default=64, help="images per GPU (BASELINE config: 64)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--objective", default="full", choices=["full", "ae"],
                    help="full: the shipped two-optimizer objective (L1+LPIPS+GAN+codebook, then the discriminator); ae: L1+codebook")
    ap.add_argument("--reuse-forward", action="store_true",
                    help="discriminator step reuses the generator step's reconstruction instead of a second autoencoder forward "
                         "(NOT the reference's schedule; reported in config.objective)")
    ap.add_argument("--no-ae-only", action="store_true", help="skip the secondary autoencoder-only measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the bf16-vs-fp32 code-index mismatch block and the block that "
                    "scores both precisions against the reference goldens")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the fp32 (parity-mode) throughput block")
    ap.add_argument("--no-vq-microbench", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python (no hipGraph replay of the step)")
    ap.add_argument("--mode", default="auto", choices=["auto", "graph", "eager"],
                    help="how the timed steps are launched: replays of the recorded step (host ~1 ms per step), eager launches "
                         "(weight gradients on a second stream; ~40-70 ms of host work per step), or -- one GPU only -- whichever "
                         "three untimed calibration steps show to be faster on this host (default)")
    ap.add_argument("--no-extras", action="store_true", help="skip the compact BASELINE config 4 / 5 blocks (bench_extra.py workloads)")
    ap.add_argument("--extras-budget", type=float, default=150.0, help="seconds the extra workloads may take in all")
    ap.add_argument("--vq-only", action="store_true", help="only the VQ-argmin micro-benchmark (kernel iteration); prints its JSON")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU check of the launch contract only: gloo ranks, barrier + timed loop of no-op steps, one JSON line; no GPU work")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return _spawn_ranks(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return _dry_run(args, world, rank)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: an MI355X is required"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = world == 1 and os.environ.get("DVQ_FORCE_DP", "0") == "1"
    if world > 1 or force_dp:
        # DVQ_FORCE_DP=1 on one GPU: a ONE-rank RCCL group, so that the data-parallel step (in-backward bucket launches, exchange
        # points cutting the recorded step into segments, waits) is what gets timed -- the cost of that machinery without a fabric
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dp:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from dynamicvectorquantization_amd import _lib
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.config import instantiate_from_config
    from dynamicvectorquantization_amd.trainer import Trainer
    _lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
    rt.set_compute_dtype(args.dtype)
    rt.set_impl(int(os.environ.get("DVQ_IMPL", "0")))     # 0 auto; 2 LDS-DMA MFMA kernels; 3 register-staged (A/B)

    if args.vq_only:
        print(json.dumps({"vq_argmin": vq_microbench(dev, reps=50)}), flush=True)
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(objective, steps, warmup, profile, no_graph=args.no_graph):
        """build the model + trainer for `objective`, W untimed + K timed steps; -> (seconds [max over ranks], host issue
        seconds, per-kernel profile of the last timed step, model)"""
        torch.manual_seed(0)       # identical initial weights on every rank
        model = instantiate_from_config(full_config(objective, args.bs)).to(dev)
        model.reuse_generator_forward = bool(args.reuse_forward)
        from dynamicvectorquantization_amd.trainer import reference_learning_rate
        from dynamicvectorquantization_amd import runtime as rt
        model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, world, args.bs)     # train.py:248-257
        model.training_steps, model.steps_per_epoch = 100000, 1000
        model.train()
        GRAPH_AFTER = 3
        trainer = Trainer(model, max_steps=steps, use_graph=not no_graph, graph_after=GRAPH_AFTER)
        # untimed initialisation before the W warmup steps: kernel code-object loading, allocator growth, packed-weight
        # tables (the first steps of a process are host-bound on these one-off costs), then the step is recorded as a
        # hipGraph (Trainer step capture) and the recording is replayed once
        # [counted eager step] + GRAPH_AFTER eager steps + [record + first replay]; two eager steps when nothing is recorded
        SETUP = GRAPH_AFTER + 2 if not no_graph else 2
        # two distinct resident batches per rank (synthetic half-flat images: fine ratio exactly 0.5)
        nb = 2
        imgs = [torch.from_numpy(synth.half_flat_images(args.bs, 256, seed=1234 + 17 * rank + 1000 * i)).to(dev) for i in range(nb)]
        batches = [{"image": im} for im in imgs]
        n_launches = 2400
        for i in range(SETUP):
            if profile and i == 0:
                K.profile_count_start()                  # size the HIP-event pool of the profiled step
            trainer.train_step(batches[i % nb], i)
            if profile and i == 0:
                n_launches = max(2400, K.profile_count_stop())
        launch_mode = "eager" if no_graph or trainer._graph is None else "graph"
        calib = None
        if args.mode == "eager" and launch_mode == "graph":
            trainer.use_graph, launch_mode = False, "eager"
        elif args.mode == "auto" and launch_mode == "graph" and world == 1 and args.dtype == "bf16":
            # (fp32 / fp32x3 steps are GPU-bound by a wide margin and their recording holds ~125 GB: no eager working set beside it)
            # untimed calibration: eager steps run the conv weight gradients on a second stream (a replay of the recorded step
            # cannot: DESIGN 3a) and win by ~2.5 % when one host core keeps up with the launches; the recorded step wins otherwise
            def probe(use_graph, n=3):
                trainer.use_graph = use_graph
                for j in range(1 if use_graph else 5):       # eager steps first re-grow the allocator pool the capture emptied
                    trainer.train_step(batches[j % nb], SETUP)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for j in range(n):
                    trainer.train_step(batches[j % nb], SETUP)
                torch.cuda.synchronize()
                return (time.perf_counter() - tp) / n
            t_graph, t_eager = probe(True), probe(False)
            calib = {"graph_ms": round(t_graph * 1e3, 2), "eager_ms": round(t_eager * 1e3, 2)}
            trainer.use_graph = t_graph <= t_eager * 1.005        # ties go to the recorded step (no dependence on the host's launch rate)
            launch_mode = "graph" if trainer.use_graph else "eager"
        for i in range(warmup):
            trainer.train_step(batches[i % nb], SETUP + i)
        if profile:
            barrier()
            K.profile_prepare(n_launches + 16)          # HIP events for ONE step, created outside the timed region
        barrier()
        t0 = time.perf_counter()
        call_s = []
        for i in range(steps):
            tc = time.perf_counter()
            trainer.train_step(batches[i % nb], SETUP + warmup + i)
            call_s.append(time.perf_counter() - tc)
        enqueue_all = time.perf_counter() - t0   # until the last step was handed to the GPU (no sync inside); includes waits for
                                                 # queue space: the runtime keeps only ~10 launches of a recorded step in flight
        # host WORK per step: calls that had to wait for queue space excluded (mean of the faster half of the calls)
        fast = sorted(call_s)[: max(1, (len(call_s) + 1) // 2)]
        host = sum(fast) / len(fast) * steps
        barrier()
        dt = time.perf_counter() - t0
        # host cost of handing ONE step to an idle device (no waiting for queue space): outside the timed region
        t_idle = time.perf_counter()
        trainer.train_step(batches[0], SETUP + warmup + steps)
        host_idle = time.perf_counter() - t_idle
        torch.cuda.synchronize()
        prof = {}
        if profile:
            # per-kernel HIP-event timing: ONE extra step right after the timed region, launched eagerly (events cannot be
            # recorded inside a graph replay) -- the same kernels on the same shapes as the replayed steps
            vq_seen = {}
            orig_fwd = model.quantize.fwd

            def spy(h, mask, tape):         # the rows / codebook the quantiser sees at this point of training (VQ micro-benchmark)
                if "x" not in vq_seen:
                    vq_seen["x"] = h.reshape(-1, h.shape[-1]).detach().clone()
                    vq_seen["cb"] = model.quantize.codebook._codebook().clone()
                return orig_fwd(h, mask, tape)
            model.quantize.fwd = spy
            # (single stream for the per-kernel events: with the weight gradients of an eager step on their side stream two kernels
            # share the chip and each one's bracket would also hold the other's time)
            prev_side = os.environ.get("DVQ_SIDE_WGRAD")
            os.environ["DVQ_SIDE_WGRAD"] = "0"
            K.profile_start()
            trainer.train_step(batches[steps % nb], SETUP + warmup + steps)
            prof = K.profile_stop()
            if prev_side is None:
                os.environ.pop("DVQ_SIDE_WGRAD", None)
            else:
                os.environ["DVQ_SIDE_WGRAD"] = prev_side
            model.quantize.fwd = orig_fwd
            model._vq_seen = (vq_seen["x"], vq_seen["cb"]) if "x" in vq_seen else None
        exposed = None
        if world > 1 or force_dp:
            # how long the compute stream sits in the gradient-exchange waits of one step (events around GradBuckets.wait())
            for gb in trainer.buckets:
                gb.measure = []
            was = trainer.use_graph
            trainer.use_graph = False
            trainer.train_step(batches[0], SETUP + warmup + steps + 1)
            torch.cuda.synchronize()
            exposed = round(sum(a.elapsed_time(b) for gb in trainer.buckets for a, b in gb.measure), 3)
            for gb in trainer.buckets:
                gb.measure = None
            trainer.use_graph = was
        graph_info = {"enabled": trainer._graph is not None, "timed_steps": launch_mode, "calibration": calib, "replays": trainer.graph_replays,
                      "allreduce_exposed_ms": exposed,
                      "host_enqueue_all_ms_per_step": round(enqueue_all / steps * 1e3, 2),
                      "segments": trainer._graph["sg"].n_segments() if trainer._graph is not None else 0,
                      "replay": rt.step_replay_mode() if trainer._graph is not None else None,
                      "host_ms_one_step_idle_queue": round(host_idle * 1e3, 2),
                      "host_call_ms_fast_half": round(host / steps * 1e3, 2),
                      "launches_main_side_waits": trainer._graph["sg"].launch_counts() if trainer._graph is not None else None,
                      "fine_ratio": float(model._logged.get("train_fine_ratio", torch.tensor(float("nan"))))}
        model._logged = {}
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        trainer.drop_graph()
        model._graph_info = graph_info
        return dt, host, prof, model

    dt_, host_issue, prof, model = run(args.objective, args.steps, args.warmup, True)
    ratio = model._graph_info.pop("fine_ratio")
    graph_info, vq_seen = model._graph_info, getattr(model, "_vq_seen", None)
    parity = None
    if rank == 0 and not args.no_parity:
        try:
            parity = parity_bf16(model, torch.from_numpy(synth.half_flat_images(args.bs, 256, seed=4321)).to(dev))
        except Exception as e:          # evidence block: never costs the headline line
            parity = {"failed": f"{type(e).__name__}: {str(e)[:160]}"}

    parity_ref = None
    if rank == 0 and not args.no_parity:
        try:
            parity_ref = parity_vs_reference(dev)
        except Exception as e:
            parity_ref = {"failed": f"{type(e).__name__}: {str(e)[:160]}"}

    ae_only = None
    if args.objective == "full" and not args.no_ae_only:
        # SURVEY 8d asks for both accountings: the autoencoder-only step (L1 + codebook) beside the complete objective
        del model
        gc.collect()              # (model <-> tape closures are reference cycles: free the first model's saved activations now)
        torch.cuda.empty_cache()
        k2 = max(2, min(4, args.steps))
        dt2, _, _, m2 = run("ae", k2, 2, False)
        ips2 = world * args.bs * k2 / dt2
        ae_only = {"value": round(ips2, 2), "unit": "images/sec", "steps": k2, "warmup": 2, "ms_per_step": round(dt2 / k2 * 1e3, 3),
                   "objective": OBJECTIVES["ae"], "step_mfma_frac": round(ips2 / world * STEP_FLOP_PER_IMG["ae"] / PEAK_BF16, 4)}
        del m2
        gc.collect()
    fp32_mode = fp32x3_mode = None
    if args.objective == "full" and args.dtype == "bf16" and world == 1 and not args.no_fp32_mode:
        # the precisions that meet north_star's tolerance (index-exact, 1e-3) get a throughput and a roofline.
        #   fp32   : every kernel instantiated for fp32 operands, products on v_mfma_f32_32x32x2_f32 (roofline: the fp32 matrix peak)
        #   fp32x3 : the same fp32 tensors, products as three bf16 MFMA passes on two-plane operands (~2^-17 per product; round 5)
        # 3 timed eager steps each (a step takes 0.6 - 1.2 s: host launch work is irrelevant, nothing is recorded)
        def precise_mode(dtype_name, peak):
            try:
                gc.collect()
                torch.cuda.empty_cache()
                rt.set_compute_dtype(dtype_name)
                k3, w3 = 3, 2              # (two untimed steps behind the two set-up steps: the caching allocator has stopped growing by then)
                dt3, _, prof3, m3 = run("full", k3, w3, True, no_graph=True)
                ips3 = world * args.bs * k3 / dt3
                fam3 = {k: v for k, v in prof3.items() if k != "vq_argmin" and v["ms"] > 0}
                dom3 = max(fam3, key=lambda k: fam3[k]["ms"]) if fam3 else None
                roof3 = None
                if dom3:
                    v = fam3[dom3]
                    ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
                    roof3 = {"kernel": f"{dom3}<{dtype_name}>", "bound": "mfma", "achieved": round(ach, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                             "frac": round(ach / (peak / 1e12), 4), "launches": v["launches"],
                             "avg_launch_ms": round(v["ms"] / max(1, v["launches"]), 4)}
                    if dtype_name == "fp32x3":
                        roof3["note"] = "algorithmic flop against the bf16 peak; the kernel issues 3 bf16 MFMA passes per product"
                out3 = {"value": round(ips3, 2), "unit": "images/sec", "steps": k3, "warmup": w3, "ms_per_step": round(dt3 / k3 * 1e3, 2),
                        "dtype": dtype_name, "launch": "eager", "roofline": roof3,
                        "kernel_families": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"], 2),
                                                "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in fam3.items()}}
                del m3
                return out3
            except Exception as e:          # evidence block: never costs the headline line
                return {"failed": f"{type(e).__name__}: {str(e)[:200]}"}
            finally:
                rt.set_compute_dtype("fp32")          # (switches the split products off again)
                rt.set_compute_dtype(args.dtype)
                gc.collect()
                torch.cuda.empty_cache()

        model = None
        fp32_mode = precise_mode("fp32", PEAK_F32)
        if "value" in fp32_mode:
            fp32_mode["step_mfma_frac_fp32_peak"] = round(fp32_mode["value"] / world * STEP_FLOP_PER_IMG["full"] / PEAK_F32, 4)
            fp32_mode["note"] = ("the complete two-optimizer step of the same workload in parity mode: every kernel instantiated for fp32 operands "
                                 "(the mode the reference goldens are met in at 1e-3 / index-exact: parity_bf16_vs_reference.fp32)")
        fp32x3_mode = precise_mode("fp32x3", PEAK_BF16)
        if "value" in fp32x3_mode:
            fp32x3_mode["note"] = ("fp32 tensors, statistics, gradients and accumulators; convolution / GEMM products as three bf16 MFMA passes on "
                                   "operands split into two bf16 planes in registers (dvq_set_fp32_split): parity_bf16_vs_reference.fp32x3")
    if rank == 0:
        ips = world * args.bs * args.steps / dt_
        fam = {k: dict(v, ms_per_launch=v["ms"] / max(1, v["launches"]),
                       TFLOPs=(v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0) for k, v in prof.items()}
        conv = [k for k in fam if k != "vq_argmin"]
        dom = max(conv, key=lambda k: fam[k]["ms"]) if conv else None      # the kernel the step spends most time in
        roofline = None
        if dom:
            v = fam[dom]
            ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
            roofline = {"kernel": dom + "<bf16>" if "igemm" in dom else dom,
                        "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / (PEAK_BF16 / 1e12), 4), "traffic": None,
                        "launches": v["launches"], "avg_launch_ms": round(v["ms_per_launch"], 4),
                        "alg_flops_per_launch": v["flops"] / max(1, v["launches"]),
                        "alg_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                        "timed": "HIP events around every launch of this kernel during one eagerly launched single-stream step right "
                                 "after the timed region; the timed steps themselves are " +
                                 (("launch-list replays of the recorded sequence (csrc/cmdlist.hip: main + side stream)" if graph_info.get("replay") == "list"
                                   else "hipGraph replays of the same launch sequence") if graph_info["timed_steps"] == "graph" else
                                  "eager launches of the same sequence with the conv weight gradients on a second stream")}
            # HBM traffic per launch: PMC counters cannot be collected from inside this process; the figure comes from the
            # committed rocprofv3 --pmc passes over this same command (tools/gpu_pmc_bench.sh -> tools/pmc_summarise.py)
            # context for `frac`: `peak` is the nominal dense bf16 rate at the maximum clock; under sustained MFMA load the chip is
            # power-limited, and how far depends on the operand bit patterns.  The same process measures what a register-only
            # MFMA loop (no LDS, no memory) sustains with random bf16 operands and with zeros.
            try:
                rnd, zer = K.probe_mfma_rate(True), K.probe_mfma_rate(False)
                roofline["sustained_mfma"] = {"random_operands_TFLOPs": round(rnd[0], 1), "random_operands_MHz": round(rnd[1]),
                                              "zero_operands_TFLOPs": round(zer[0], 1), "zero_operands_MHz": round(zer[1]),
                                              "frac_of_random_operand_rate": round(ach / max(rnd[0], 1e-9), 4)}
            except Exception as e:      # diagnostics only
                roofline["sustained_mfma"] = {"error": str(e)}
            pmc = _pmc_traffic(dom)
            if pmc is not None and args.objective == "full" and args.bs == 64:
                roofline["traffic"] = pmc["hbm_bytes_per_launch"]
                roofline["traffic_source"] = pmc["source"]
        out = {
            "metric": METRIC, "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_ / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DQ-VAE dual F=16/8 (configs/stage1/dqvae-entropy-dual-r05_imagenet.yml), codebook 1024x256, "
                                   f"bs={args.bs}/GPU, 256x256 half-flat synthetic images",
                       "objective": OBJECTIVES[args.objective] + (" [discriminator step reuses the generator step's reconstruction]"
                                                                         if args.reuse_forward else ""),
                       "global_batch": world * args.bs, "parallelism": f"dp{world}", "fine_ratio": ratio,
                       "step_graph": graph_info},
            "step_mfma_frac": round(ips / world * STEP_FLOP_PER_IMG[args.objective] / PEAK_BF16, 4),
            "step_flop_per_img": STEP_FLOP_PER_IMG[args.objective],
            # host WORK to hand one step to an idle device; the mean of the faster half of the timed calls (which still contains waits
            # for queue space: two steps of 2200 packets do not fit the queues) is config.step_graph.host_call_ms_fast_half
            "host_issue_ms_per_step": graph_info.get("host_ms_one_step_idle_queue"),
            "rccl_ranks": dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0,
            "allreduce_exposed_ms": graph_info.get("allreduce_exposed_ms"),
            "roofline": roofline,
            "kernel_families": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"], 3),
                                    "TFLOPs": round(v["TFLOPs"], 2)} for k, v in fam.items()},
        }
        out["ae_only"] = ae_only
        out["parity_bf16"] = parity
        out["parity_bf16_vs_reference"] = parity_ref
        out["fp32_mode"] = fp32_mode
        out["fp32x3_mode"] = fp32x3_mode
        if not args.no_vq_microbench:
            out["vq_argmin"] = vq_microbench(dev, in_training=vq_seen)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.objective)
        if world == 1 and not args.no_extras and args.objective == "full" and args.bs == 64:
            # BASELINE configs 4 / 5 in the same driver command (compact blocks; bench_extra.py prints the full lines): each runs in
            # its own process after this one has released its memory, inside a time budget
            model = None                           # (already released when the autoencoder-only measurement ran)
            torch.cuda.empty_cache()
            out["extra_workloads"] = extra_workloads(args.extras_budget)
        try:        # RCCL prints its version banner through C stdio: flush it NOW so that the JSON line is the last line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
