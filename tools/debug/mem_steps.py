#!/usr/bin/env python3
"""allocated / reserved device memory per eager training step (bs 64, complete objective): python tools/debug/mem_steps.py [dtype] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dynamicvectorquantization_amd import runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer
dev = torch.device("cuda", 0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rt.set_compute_dtype(dtype)
torch.manual_seed(0)
model = instantiate_from_config(bench.full_config("full", 64)).to(dev)
model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-5, 100000, 1000
model.train()
tr = Trainer(model, max_steps=steps, use_graph=False)
batches = [{"image": torch.from_numpy(synth.half_flat_images(64, 256, seed=900 + i)).to(dev)} for i in range(2)]
for i in range(steps):
    torch.cuda.reset_peak_memory_stats()
    out = tr.train_step(batches[i % 2], i)
    torch.cuda.synchronize()
    print(f"step {i}: allocated {torch.cuda.memory_allocated() / 2**30:7.2f} GiB  peak {torch.cuda.max_memory_allocated() / 2**30:7.2f} GiB  "
          f"reserved {torch.cuda.memory_reserved() / 2**30:7.2f} GiB  losses {[round(float(l), 4) for l in out]}", flush=True)
