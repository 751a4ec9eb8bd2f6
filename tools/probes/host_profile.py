import os, sys, cProfile, pstats, io, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt, synth
from dynamicvectorquantization_amd.config import instantiate_from_config
from dynamicvectorquantization_amd.trainer import Trainer
import bench
dev = torch.device("cuda:0")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
model = instantiate_from_config(bench.full_config()).to(dev)
model.learning_rate = 1e-4; model.training_steps, model.steps_per_epoch = 1000, 100
model.train()
tr = Trainer(model, max_steps=10)
bs = int(os.environ.get("BS", 64))
x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=1)).to(dev)
for i in range(2):
    tr.train_step({"image": x}, i)
torch.cuda.synchronize()
t0 = time.perf_counter(); tr.train_step({"image": x}, 2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"plain: issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
pr = cProfile.Profile(); pr.enable()
tr.train_step({"image": x}, 3)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
