#!/usr/bin/env python3
"""Does the bf16 benchmark path TRAIN like the fp32 (tolerance-meeting) path?  The same model (seed 0), the same batches, the complete
two-optimizer objective of the shipped YAML at bs 64, 256 x 256: N steps through both instantiations of the kernels; per step the
autoencoder loss and the discriminator loss of each precision and their relative difference, plus how far the parameters are apart at
the end.  (Rounding differences are expected to be amplified by Adam -- its first steps move every weight by +-lr whatever the gradient
magnitude -- so the measure is the LOSS curves, not bit patterns.)

    python tools/debug/r5_loss_tracking.py [steps]   ->  gpurun_out/r5_loss_tracking.json
"""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from dynamicvectorquantization_amd import _lib, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer, reference_learning_rate

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BS = 64
batches = [{"image": torch.from_numpy(synth.half_flat_images(BS, 256, seed=900 + i)).to(dev)} for i in range(4)]


def run(tag, dtype, warmup, perturb=0.0, steps=STEPS):
    rt.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = instantiate_from_config(bench.full_config("full", BS)).to(dev)
    model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, 1, BS)
    model.training_steps, model.steps_per_epoch = 100000, 1000
    if not warmup:
        model.warmup_epochs = 0             # full learning rate from step 0: the harder case for the comparison
    model.train()
    tr = Trainer(model, max_steps=steps, use_graph=False)
    t0 = time.time()
    rows = []
    for i in range(steps):
        b = batches[i % 4]
        if perturb:                          # control: the same run with the images moved by `perturb` x N(0,1) -- what chaos does to an exact run
            g = torch.Generator(device=dev).manual_seed(77 + i)
            b = {"image": b["image"] + perturb * torch.randn(b["image"].shape, device=dev, generator=g)}
        out = tr.train_step(b, i)
        rows.append([float(l) for l in out])
    torch.cuda.synchronize()
    print(f"{tag}: {steps} steps in {time.time() - t0:.1f} s; last losses {rows[-1]}", flush=True)
    fin = {n: p.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad and not n.startswith("loss.perceptual")}
    del tr, model
    gc.collect()                 # (the model <-> tape closures form reference cycles: without this the next run starts with the last tape resident)
    torch.cuda.empty_cache()
    return rows, fin


def compare(a, b):
    (ca, fa), (cb, fb) = a, b
    rel = [[abs(x - y) / max(1e-12, abs(y)) for x, y in zip(ra, rb)] for ra, rb in zip(ca, cb)]
    num = sum(float((fa[n] - fb[n]).double().pow(2).sum()) for n in fb)
    den = sum(float(fb[n].double().pow(2).sum()) for n in fb)
    return {"rel_diff_per_step": [[round(v, 5) for v in r] for r in rel], "step0_rel_diff": [round(v, 5) for v in rel[0]],
            "max_rel_diff_aeloss": round(max(r[0] for r in rel), 4), "mean_rel_diff_aeloss": round(sum(r[0] for r in rel) / len(rel), 4),
            "max_rel_diff_discloss": round(max(r[-1] for r in rel), 4), "mean_rel_diff_discloss": round(sum(r[-1] for r in rel) / len(rel), 4),
            "params_rel_l2_distance_at_end": round((num / den) ** 0.5, 5)}


res = {"steps": STEPS, "bs": BS, "objective": "complete two-optimizer step, shipped YAML"}
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
OUT = os.path.join(bench.REPO, "gpurun_out", "r5_loss_tracking_B.json" if ONLY == "B" else "r5_loss_tracking.json")
if ONLY != "B":
    # A: no LR warm-up (every Adam step moves every weight by ~lr): bf16 vs fp32, and -- the control -- fp32 vs fp32 with the input images
    #    moved by 1e-6 x N(0,1) (far below one 8-bit grey level): how far do two CORRECT runs drift apart?
    a_bf16, a_fp32, a_ctrl = run("A bf16", "bf16", False), run("A fp32", "fp32", False), run("A fp32 (images + 1e-6 noise)", "fp32", False, 1e-6)
    res["no_warmup"] = {"losses_bf16": a_bf16[0], "losses_fp32": a_fp32[0], "losses_fp32_control": a_ctrl[0],
                        "bf16_vs_fp32": compare(a_bf16, a_fp32), "fp32_control_vs_fp32": compare(a_ctrl, a_fp32)}
    del a_bf16, a_fp32, a_ctrl
    os.makedirs(os.path.join(bench.REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(bench.REPO, "gpurun_out", "r5_loss_tracking.json"), "w"), indent=1)
    for kk, v in res["no_warmup"].items():
        if isinstance(v, dict):
            print("no_warmup", kk, {x: v[x] for x in v if x != "rel_diff_per_step"}, flush=True)
# B: the YAML's own warm-up (lr ramps from ~0): the trajectories stay together and the difference is the forward / backward precision
b_bf16, b_fp32 = run("B bf16", "bf16", True, steps=8), run("B fp32", "fp32", True, steps=8)
res["yaml_warmup"] = {"losses_bf16": b_bf16[0], "losses_fp32": b_fp32[0], "bf16_vs_fp32": compare(b_bf16, b_fp32)}
os.makedirs(os.path.join(bench.REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(OUT, "w"), indent=1)
for k in ("no_warmup", "yaml_warmup"):
    for kk, v in res.get(k, {}).items():
        if isinstance(v, dict):
            print(k, kk, {x: v[x] for x in v if x != "rel_diff_per_step"})
