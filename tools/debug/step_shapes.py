#!/usr/bin/env python3
"""Per-geometry conv timings of ONE headline train step (DVQ_PROFILE_SHAPES=1): which shapes the implicit-GEMM / halo kernel
families spend their time on.  HIP events around every launch; prints rows sorted by time.  DVQ_SHAPES_DTYPE=bf16|fp32|fp32x3."""
import os, sys
os.environ["DVQ_PROFILE_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dynamicvectorquantization_amd import _lib, kernels as K, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype(os.environ.get("DVQ_SHAPES_DTYPE", "bf16"))
torch.manual_seed(0)
bs = 64
model = instantiate_from_config(bench.full_config("full")).to(dev)
model.learning_rate, model.training_steps, model.steps_per_epoch = 4.5e-6 * bs, 100000, 1000
model.train()
tr = Trainer(model, max_steps=100)
batches = [{"image": torch.from_numpy(synth.half_flat_images(bs, 256, seed=5 + i)).to(dev)} for i in range(2)]
for i in range(3):
    tr.train_step(batches[i % 2], i)
torch.cuda.synchronize()
K.profile_start(4000)
tr.train_step(batches[1], 3)
prof = K.profile_stop()
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"total timed {tot:.1f} ms")
for name, v in rows[:int(os.environ.get("TOP", "45"))]:
    print(f"{v['ms']:8.3f} ms  x{v['launches']:3d}  {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.0f} TF/s  {name}")
