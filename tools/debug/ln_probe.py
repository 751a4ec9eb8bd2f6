"""LayerNorm backward at the stage-2 shape [20736, 1024] bf16: time per call (DVQ_LN_BWD_WAVES sweeps the wave count)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K
dev = torch.device("cuda:0")
rows, c = 20736, 1024
x = torch.randn(rows, c, device=dev).to(torch.bfloat16); dy = torch.randn(rows, c, device=dev).to(torch.bfloat16)
g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
y, mr = K.layernorm_fwd(x, g, b, 1e-5, want_stats=True)
dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
for _ in range(3): K.layernorm_bwd(x, dy, mr, g, dg, db, dy)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): K.layernorm_bwd(x, dy, mr, g, dg, db, dy)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f"DVQ_LN_BWD_WAVES={os.environ.get('DVQ_LN_BWD_WAVES','2048')}: {ms*1e3:.1f} us per call = {4 * rows * c * 2 / ms / 1e9:.2f} TB/s (x, dy, dres in, dx out)")
