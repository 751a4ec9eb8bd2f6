"""shader-clock stamps of one wave of the bf16 VQ main kernel (DVQ_VQ_DBG=32 complete / 61 MFMAs only): where its cycles go"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamicvectorquantization_amd import kernels as K, synth
dev = torch.device("cuda:0")
x, cb = synth.vq_inputs(65536, 256, 1024, "normal", 0)
xt = torch.from_numpy(x).to(dev).to(torch.bfloat16); cbt = torch.from_numpy(cb).to(dev)
prep = K.vq_prepare(cbt)
for _ in range(5):
    K.vq_argmin(xt, cbt, prep, impl=2)
torch.cuda.synchronize()
ws = next(iter(K._vq_ws.values()))
t = ws[64:104].view(torch.int64).cpu().numpy()          # VqWs.pad[9 ..]: five 64-bit stamps
d = [int(t[i + 1] - t[i]) for i in range(4)]
print(f"DVQ_VQ_DBG={os.environ.get('DVQ_VQ_DBG')}: rows+norms {d[0]}  first stage wait {d[1]}  stage loop {d[2]}  select {d[3]}  total {int(t[4]-t[0])} shader cycles")
