import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (mred, i, j) in [(20576, 1024, 1024), (20576, 3072, 1024), (20576, 4096, 1024), (20576, 1024, 4096), (65536, 256, 256), (16384, 512, 512)]:
    a2 = (torch.rand(mred, i, device=dev)*2-1).to(torch.bfloat16)
    b2 = (torch.rand(mred, j, device=dev)*2-1).to(torch.bfloat16)
    out = torch.zeros(i * j, device=dev, dtype=torch.float32)
    ms = timeit(lambda: K.gemm_tn(a2.reshape(-1), b2.reshape(-1), mred, i, j, i, j, j, out=out))
    mt = timeit(lambda: torch.matmul(a2.t(), b2))
    # transposes + NT
    def via_nt():
        at = a2.t().contiguous(); bt = b2.t().contiguous()
        return at, bt
    mtr = timeit(via_nt)
    f = 2.0 * mred * i * j
    print(f"TN Mred={mred} I={i} J={j}: ours {ms:7.3f} ms {f/ms/1e9:6.0f} TF/s   torch {mt:7.3f} ms {f/mt/1e9:6.0f} TF/s   torch-transposes {mtr:7.3f} ms", flush=True)
