import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dynamicvectorquantization_amd import synth, kernels as K, runtime as rt
from test_gpu_model import build
from dynamicvectorquantization_amd.trainer import Trainer
dev = torch.device("cuda:0")
orig = K.conv2d_dgrad
def dbg(d, dy, wt, mask=None, mask_act=0):
    print("dgrad N=%d H=%d W=%d Cin=%d Cout=%d OH=%d dy=%s wt=%s mask=%s" % (d.N, d.H, d.W, d.Cin, d.Cout, d.OH, tuple(dy.shape), tuple(wt.shape), None if mask is None else (tuple(mask.shape), mask.is_contiguous(), mask.storage_offset())), flush=True)
    r = orig(d, dy, wt, mask=mask, mask_act=mask_act)
    torch.cuda.synchronize()
    return r
K.conv2d_dgrad = dbg
with rt.compute_dtype_ctx(torch.bfloat16):
    model, _ = build("small", dev, "spread", loss="full")
    model.learning_rate, model.training_steps, model.steps_per_epoch = 1e-3, 100, 10
    model.train()
    x = torch.from_numpy(synth.half_flat_images(int(os.environ.get("BS", 2)), 64, seed=7)).to(dev)
    tr = Trainer(model, max_steps=2)
    tr.train_step({"image": x}, 0)
    torch.cuda.synchronize()
    print("step ok")
