"""Which Python lines issue device-to-device hipMemcpyAsync inside the training step?  (A recorded step keeps them as 1-D memcpy
nodes, whose parameters a launch list cannot read back on ROCm 7.2: the step has to use kernel copies.)  Profiles ONE eager step
with Python stacks and lists every aten::copy_ / aten::clone with the package frames above it."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dynamicvectorquantization_amd import runtime as rt, synth  # noqa: E402
from test_gpu_stepgraph import _make  # noqa: E402

dev = torch.device("cuda:0")
for loss in ("ae", "full"):
    xs = [torch.from_numpy(synth.half_flat_images(2, 64, seed=40 + i)).to(dev) for i in range(3)]
    with rt.compute_dtype_ctx(torch.bfloat16 if os.environ.get("BF16") else torch.float32):
        model, tr = _make(dev, False, loss)
        for i in range(2):
            tr.train_step({"image": xs[i % 3]}, i)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
            tr.train_step({"image": xs[2]}, 2)
            torch.cuda.synchronize()
    print("==", loss, flush=True)
    seen = {}
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous") or "emcpy" in ev.name:
            st = [s for s in (ev.stack or []) if "dynamicvectorquantization_amd" in s or "autograd" in s][:3]
            key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:60] for s in st))
            seen[key] = seen.get(key, 0) + 1
    for k, v in sorted(seen.items()):
        print("  ", v, k, flush=True)
