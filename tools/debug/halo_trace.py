#!/usr/bin/env python3
"""DVQ_HALO_DBG=6: where do the main loops and epilogues of the 3x3 halo conv's workgroups fall in time, per CU?  Prints, for a few
CUs, the [start, main-loop end, end] intervals of the workgroups that ran there, and chip-wide how much of the epilogue time of a
workgroup overlaps a main loop of its CU neighbour."""
import os, sys
os.environ["DVQ_HALO_DBG"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamicvectorquantization_amd import kernels as K, runtime as rt, _lib
from dynamicvectorquantization_amd.layers import Conv2d
dev = torch.device("cuda:0")
rt.set_compute_dtype(torch.bfloat16)
B, H, C = 64, 256, 128
conv = Conv2d(C, C, 3, 1, 1).to(dev)
w, wt, bias = conv.packed(torch.bfloat16)
g = torch.randn(B, H, H, C, device=dev)
x = (g * torch.sigmoid(g)).to(torch.bfloat16)
r = g.to(torch.bfloat16)
d = conv._desc(x)
K.ensure_workspace(dev)
stats = torch.zeros(B, 32, 2, dtype=torch.float64, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
for _ in range(3):
    y = K.conv2d_fwd(d, x, w, bias, r if "res" in mode else None, out_stats=stats if "stats" in mode else None, out_groups=32 if "stats" in mode else 0)
torch.cuda.synchronize()
nrec = 16384
buf = np.zeros((nrec, 6), dtype=np.uint64)
_lib.check(_lib.load().dvq_halo_trace_read(buf.ctypes.data, nrec), "trace")
t0 = buf[:, 1].min()
t4 = ((buf[:, 0] >> np.uint64(16)) - (t0 & np.uint64((1 << 48) - 1))).astype(np.float64) / 100
st, sd = (buf[:, 4] - t0).astype(np.float64) / 100, (buf[:, 5] - t0).astype(np.float64) / 100
key, s, m, e = (buf[:, 0] & np.uint64(0xffff)).astype(np.int64), (buf[:, 1] - t0).astype(np.float64) / 100, (buf[:, 2] - t0).astype(np.float64) / 100, (buf[:, 3] - t0).astype(np.float64) / 100
print(f"mode {mode}: {nrec} workgroups on {len(np.unique(key))} CUs, kernel span {e.max():.1f} us; mean main loop {np.mean(m - s):.2f} us, mean epilogue {np.mean(e - m):.2f} us")
ov_tot, ep_tot = 0.0, 0.0
for k in np.unique(key):
    idx = np.where(key == k)[0]
    for i in idx:
        ep_tot += e[i] - m[i]
        for j in idx:
            if j != i:
                ov_tot += max(0.0, min(e[i], m[j]) - max(m[i], s[j]))
print(f"epilogue split: staging {np.mean(st - m):.2f} us, store loop {np.mean(sd - st):.2f} us, statistics {np.mean(t4 - sd):.2f} us, store drain {np.mean(e - t4):.2f} us")
print(f"epilogue time overlapped by a CU neighbour's main loop: {100 * ov_tot / ep_tot:.1f} %")
for k in np.unique(key)[:3]:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(s[idx])][:10]
    print(f"CU {k:#x}: " + "  ".join(f"[{s[i]:.1f} {m[i]:.1f} {e[i]:.1f}]" for i in idx))
