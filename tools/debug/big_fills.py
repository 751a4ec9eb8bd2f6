#!/usr/bin/env python3
"""Which host call sites issue large fills during a stage-2 train step?  Wraps torch.zeros / zeros_like / Tensor.zero_ / fill_ and prints
every fill of more than 32 MB with its caller (run: python tools/debug/big_fills.py)."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
seen = collections.Counter()


def note(nbytes, kind):
    if nbytes < (32 << 20):
        return
    fr = [f for f in traceback.extract_stack()[:-2] if "dynamicvectorquantization_amd" in f.filename or "bench_extra" in f.filename]
    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:])
    seen[(kind, nbytes >> 20, where)] += 1


_zeros, _zeros_like, _zero_, _fill_, _new_zeros = torch.zeros, torch.zeros_like, torch.Tensor.zero_, torch.Tensor.fill_, torch.Tensor.new_zeros


def zeros(*a, **k):
    t = _zeros(*a, **k)
    note(t.numel() * t.element_size(), "zeros")
    return t


def zeros_like(x, *a, **k):
    note(x.numel() * x.element_size(), "zeros_like")
    return _zeros_like(x, *a, **k)


def zero_(self):
    note(self.numel() * self.element_size(), "zero_")
    return _zero_(self)


def fill_(self, v):
    note(self.numel() * self.element_size(), "fill_")
    return _fill_(self, v)


torch.zeros, torch.zeros_like, torch.Tensor.zero_, torch.Tensor.fill_ = zeros, zeros_like, zero_, fill_
sys.argv = ["bench_extra.py", "--workload", "stage2"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench_extra.py"), run_name="__main__")
for (kind, mb, where), n in seen.most_common(20):
    print(f"{n:4d} x {mb:6d} MB {kind:10s} {where}")
