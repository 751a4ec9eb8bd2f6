#!/usr/bin/env python3
"""Phase time stamps of workgroup 0 inside dvq_decode_stack (DVQ_DECODE_TRACE=1): p6c18 widths, batch 8, 18 blocks, cache row 600."""
import os, sys
os.environ["DVQ_DECODE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
from dynamicvectorquantization_amd import kernels as K, _lib
from dynamicvectorquantization_amd._lib import DecodeLayer
dev = torch.device("cuda:0")
B, C, NH, F, TMAX, NL = int(os.environ.get("B", 8)), 1024, 16, 4096, 1300, int(os.environ.get("NL", 18))
bf = lambda *s: (torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)
keep, arr = [], (DecodeLayer * NL)()
for i in range(NL):
    ws = [bf(C, C), bf(C, C), bf(C, C), bf(C, C), bf(F, C), bf(C, F)]
    bs = [torch.zeros(n, device=dev) for n in (C, C, C, C, F, C)]
    ln = [torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)]
    kc, vc = bf(B, TMAX, C), bf(B, TMAX, C)
    keep += [ws, bs, ln, kc, vc]
    arr[i] = DecodeLayer(*[t.data_ptr() for t in ws], *[t.data_ptr() for t in bs], *[t.data_ptr() for t in ln], kc.data_ptr(), vc.data_ptr())
table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
scratch = K.decode_stack_scratch(B, C, F, dev)
t_dev = torch.full((1,), int(os.environ.get("ROW", 600)), dtype=torch.long, device=dev)
x = bf(B, C)
nwg = int(os.environ.get("DVQ_DECODE_WGS", "0"))
for _ in range(3):
    K.decode_stack(table, NL, x, NH, F, TMAX, t_dev, 1e-5, scratch, nwg, table_host=arr)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    K.decode_stack(table, NL, x, NH, F, TMAX, t_dev, 1e-5, scratch, nwg, table_host=arr)
e.record(); torch.cuda.synchronize()
print(f"{NL} blocks: {s.elapsed_time(e) / 20 * 1e3:.1f} us per launch = {s.elapsed_time(e) / 20 / NL * 1e3:.1f} us per block")
off = ((4 * B * C + B * F) * 2 + 15) // 16 * 16
words = scratch[off:off + 32].view(torch.int32).cpu().numpy()
print("sync words", words[:3])
tr = scratch[off + 32:off + 32 + 1024].view(torch.int64).cpu().numpy()
names = ["start", "LN1", "qkv+stores", "barrier1", "attn+stores", "barrier2", "proj+stores", "barrier3", "LN2", "fc+stores", "barrier4", "proj2+stores", "barrier5"]
per_launch_us = s.elapsed_time(e) / 20 * 1e3
span1 = float(tr[1 + 2 * 12 - 1] - tr[1 + 12 - 1])            # block 1, in counter units
unit = per_launch_us / NL / span1                             # us per counter unit (block 1 taken as a typical block)
for blk in range(2):
    base = 1 + blk * 12
    prev = tr[base - 1]
    row = []
    for j in range(12):
        v = tr[base + j]
        row.append(f"{names[1 + j]} {(v - prev) * unit:.1f}")
        prev = v
    print(f"block {blk} (us): " + "  ".join(row))
