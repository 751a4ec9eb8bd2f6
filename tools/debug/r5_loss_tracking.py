#!/usr/bin/env python3
"""Does the bf16 benchmark path TRAIN like the fp32 (tolerance-meeting) path?  The same model (seed 0), the same batches, the complete
two-optimizer objective of the shipped YAML at bs 64, 256 x 256: N steps through both instantiations of the kernels; per step the
autoencoder loss and the discriminator loss of each precision and their relative difference, plus how far the parameters are apart at
the end.  (Rounding differences are expected to be amplified by Adam -- its first steps move every weight by +-lr whatever the gradient
magnitude -- so the measure is the LOSS curves, not bit patterns.)

    python tools/debug/r5_loss_tracking.py [steps]   ->  gpurun_out/r5_loss_tracking.json
"""
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
curves, finals = {}, {}
for tag in ("bf16", "fp32"):
    rt.set_compute_dtype(tag)
    torch.manual_seed(0)
    model = instantiate_from_config(bench.full_config("full", BS)).to(dev)
    model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, 1, BS)
    model.training_steps, model.steps_per_epoch = 100000, 1000
    model.warmup_epochs = 0                 # full learning rate from step 0: the harder case for the comparison
    model.train()
    tr = Trainer(model, max_steps=STEPS, use_graph=False)
    t0 = time.time()
    rows = []
    for i in range(STEPS):
        out = tr.train_step(batches[i % 4], i)
        rows.append([float(l) for l in out])
    torch.cuda.synchronize()
    print(f"{tag}: {STEPS} steps in {time.time() - t0:.1f} s; last losses {rows[-1]}", flush=True)
    curves[tag] = rows
    finals[tag] = {n: p.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad and not n.startswith("loss.perceptual")}
    del tr, model
    torch.cuda.empty_cache()
rel = [[abs(a - b) / max(1e-12, abs(b)) for a, b in zip(ra, rb)] for ra, rb in zip(curves["bf16"], curves["fp32"])]
num = sum(float((finals["bf16"][n] - finals["fp32"][n]).double().pow(2).sum()) for n in finals["fp32"])
den = sum(float(finals["fp32"][n].double().pow(2).sum()) for n in finals["fp32"])
res = {"steps": STEPS, "bs": BS, "objective": "complete two-optimizer step, shipped YAML, no LR warm-up",
       "losses_bf16": curves["bf16"], "losses_fp32": curves["fp32"], "rel_diff_per_step": rel,
       "max_rel_diff_aeloss": max(r[0] for r in rel), "max_rel_diff_discloss": max(r[-1] for r in rel),
       "mean_rel_diff_aeloss": sum(r[0] for r in rel) / len(rel), "params_rel_l2_distance_at_end": (num / den) ** 0.5}
os.makedirs(os.path.join(bench.REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(bench.REPO, "gpurun_out", "r5_loss_tracking.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("max_rel_diff_aeloss", "mean_rel_diff_aeloss", "max_rel_diff_discloss", "params_rel_l2_distance_at_end")}))
for i, (a, b, r) in enumerate(zip(curves["bf16"], curves["fp32"], rel)):
    print(i, [round(v, 5) for v in a], [round(v, 5) for v in b], [round(v, 4) for v in r])
