#!/usr/bin/env python3
"""cProfile of one StackGPT p6c18 train step (bs 32): where do the ~65 ms of host work per step go?"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import _lib, config as cfg, runtime as rt, synth
from dynamicvectorquantization_amd.trainer import Trainer
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(REPO)
dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
model = cfg.instantiate_from_config(cfg.load_yaml("configs/stage2/uncond_imagenet_p6c18.yml").model).to(dev)
model.learning_rate, model.min_learning_rate, model.training_steps, model.steps_per_epoch = 5e-4, 0.0, 100000, 1000
model.train()
tr = Trainer(model, max_steps=10)
batches = [{"image": torch.from_numpy(synth.half_flat_images(32, 256, seed=177 + i)).to(dev)} for i in range(2)]
for i in range(4):
    tr.train_step(batches[i % 2], i)
torch.cuda.synchronize()
t0 = time.perf_counter(); tr.train_step(batches[0], 4); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"plain: issue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile(); pr.enable()
tr.train_step(batches[1], 5)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35); print(s.getvalue()[:7000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
