import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from dynamicvectorquantization_amd import runtime as rt
from dynamicvectorquantization_amd.layers import Conv2d
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (cin, cout, k, h, w, n, mode) in [(64, 64, 1, 8, 8, 2, "rand"), (64, 64, 1, 8, 8, 2, "pix"), (32, 64, 3, 8, 8, 2, "rand")]:
    mod = Conv2d(cin, cout, k, 1, (k - 1) // 2).to(dev)
    x = torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).float()
    if mode == "rand":
        go = torch.randn(n, cout, h, w, device=dev).to(torch.bfloat16).float()
    else:   # value = pixel index (same for all channels): reveals which pixels are summed
        go = torch.arange(n * h * w, device=dev).float().view(n, 1, h, w).expand(n, cout, h, w).contiguous()
    with rt.compute_dtype_ctx(torch.bfloat16), rt.impl_ctx(2):
        xt = x.clone().requires_grad_(True)
        y = mod(xt)
        (y.float() * go).sum().backward()
    got = mod.bias.grad.cpu().numpy(); ref = go.sum(dim=(0, 2, 3)).cpu().numpy()
    print(mode, k, "got", got[:8], "\n   ref", ref[:8], "\n   ratio", (got / ref)[:8])
