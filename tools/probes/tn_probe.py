"""TN GEMM (weight-gradient shapes of the StackGPT Linear layers): the automatic route (8-phase main loop; DVQ_TN_8PHASE=0 = the
per-stage-drain kernel) against torch.matmul (hipBLASLt) as a yardstick only; full-tensor check against an fp32 product, column sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
K.ensure_workspace(dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
print("DVQ_TN_8PHASE =", os.environ.get("DVQ_TN_8PHASE", "1 (default)"))
for (mred, i, j) in [(20576, 1024, 1024), (20576, 3072, 1024), (20576, 4096, 1024), (20576, 1024, 4096), (65536, 256, 256), (16384, 512, 512), (5000, 264, 520)]:
    torch.manual_seed(mred + i)
    a2 = (torch.rand(mred, i, device=dev)*2-1).to(torch.bfloat16)
    b2 = (torch.rand(mred, j, device=dev)*2-1).to(torch.bfloat16)
    out = torch.zeros(i * j, device=dev, dtype=torch.float32)
    ms = timeit(lambda: K.gemm_tn(a2.reshape(-1), b2.reshape(-1), mred, i, j, i, j, j, out=out))
    mt = timeit(lambda: torch.matmul(a2.t(), b2))
    out.zero_()
    cs = torch.zeros(i, device=dev, dtype=torch.float32)
    K.gemm_tn(a2.reshape(-1), b2.reshape(-1), mred, i, j, i, j, j, out=out, colsum=cs)
    ref = torch.matmul(a2.float().t(), b2.float())
    err = float((out.view(i, j) - ref).abs().max() / ref.abs().max())
    cerr = float((cs - a2.float().sum(0)).abs().max() / a2.float().sum(0).abs().max())
    outs = []
    for _ in range(5):
        o = torch.zeros(i * j, device=dev, dtype=torch.float32)
        K.gemm_tn(a2.reshape(-1), b2.reshape(-1), mred, i, j, i, j, j, out=o)
        outs.append(o)
    rep = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
    f = 2.0 * mred * i * j
    print(f"TN Mred={mred} I={i} J={j}: ours {ms:7.3f} ms {f/ms/1e9:6.0f} TF/s   torch {mt:7.3f} ms {f/mt/1e9:6.0f} TF/s   err {err:.1e} colsum err {cerr:.1e} differing repeats {rep}/4", flush=True)
