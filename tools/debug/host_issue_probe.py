#!/usr/bin/env python3
"""How far ahead of the GPU does the host run?  Per-step enqueue time of the headline train step (no sync inside the loop),
plus torch's sync-debug warnings for implicit host<->device synchronisations inside a step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dynamicvectorquantization_amd import _lib, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
bs = int(os.environ.get("BS", "64"))
if os.environ.get("WORKLOAD", "stage1") == "stage2":
    from dynamicvectorquantization_amd import config as cfg
    os.chdir(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    bs = int(os.environ.get("BS", "32"))
    model = cfg.instantiate_from_config(cfg.load_yaml("configs/stage2/uncond_imagenet_p6c18.yml").model).to(dev)
    model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 5e-4, 0.0, 100000, 1000
else:
    model = instantiate_from_config(bench.full_config("full")).to(dev)
    model.learning_rate, model.training_steps, model.steps_per_epoch = 4.5e-6 * bs, 100000, 1000
model.train()
tr = Trainer(model, max_steps=100)
batches = [{"image": torch.from_numpy(synth.half_flat_images(bs, 256, seed=5 + i)).to(dev)} for i in range(2)]
for i in range(4):
    tr.train_step(batches[i % 2], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(6):
    a = time.perf_counter()
    tr.train_step(batches[i % 2], 4 + i)
    marks.append(time.perf_counter() - a)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host enqueue ms per step:", [round(m * 1e3, 1) for m in marks], "total host", round(host * 1e3, 1), "wall", round(tot * 1e3, 1))
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    tr.train_step(batches[0], 10)
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter((str(x.filename).split("/")[-1], x.lineno) for x in w)
print("implicit syncs in one step:", sum(c.values()))
for k, v in c.most_common(20):
    print("  ", k, v)
if os.environ.get("PYPROF"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(2):
        tr.train_step(batches[i % 2], 11 + i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
