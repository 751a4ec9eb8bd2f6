#!/usr/bin/env python3
"""cProfile of one KV-cached sampling run (bs 8, p6c18, random weights): where does the host time of a token step go?"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import _lib, config as cfg, runtime as rt, synth
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(REPO)
dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
model = cfg.instantiate_from_config(cfg.load_yaml("configs/stage2/uncond_imagenet_p6c18.yml").model).to(dev)
model.eval()
bs = int(os.environ.get("BS", "8"))
x = torch.from_numpy(synth.half_flat_images(bs, 256, seed=277)).to(dev)
with torch.no_grad():
    c = model.encode_to_c(x)
    kw = dict(sample=True, top_k=300, top_k_pos=100, process=False, fix_fine_position=True)
    model.sample_from_scratch(*c, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = model.sample_from_scratch(*c, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = int(r[0].shape[1] + r[1].shape[1])
    print(f"{n} token steps in {dt:.3f} s -> {dt / n * 1e3:.2f} ms per step")
    pr = cProfile.Profile()
    pr.enable()
    model.sample_from_scratch(*c, **kw)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
