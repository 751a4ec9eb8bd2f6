"""Oracle: dual-grain code map <-> sequence permutation (numpy, integer, sequential like the reference).
TEST INFRASTRUCTURE -- never imported by the product package.

Follows (behaviour, not code) /root/reference/modules/dynamic_modules/permuter.py:50-135: `forward` builds the
EOS-terminated / PAD-filled coarse and fine content + position rows ("region-first": fine cells one after the other,
"row-first": raster order of the fine grid), `forward_back` scatters them back sequentially.
"""
from __future__ import annotations

import numpy as np


def forward(indices, grain, hw1, hw2, order="region-first", content_pad=1024, content_eos=1025, cpos_pad=256, cpos_eos=257,
            fpos_pad=1024, fpos_eos=1025):
    indices, grain = np.asarray(indices, dtype=np.int64), np.asarray(grain, dtype=np.int64)
    b = indices.shape[0]
    fhw = hw1 * hw2
    rows = {"cc": [], "cp": [], "fc": [], "fp": []}
    for i in range(b):
        cc, cp, fc, fp = [], [], [], []
        for h1 in range(hw1):
            for w1 in range(hw1):
                if grain[i, h1, w1] == 0:
                    cc.append(indices[i, h1 * hw2, w1 * hw2])
                    cp.append(h1 * hw1 + w1)
                elif grain[i, h1, w1] == 1 and order == "region-first":
                    for h2 in range(hw2):
                        for w2 in range(hw2):
                            fc.append(indices[i, h1 * hw2 + h2, w1 * hw2 + w2])
                            fp.append((h1 * hw2 + h2) * fhw + w1 * hw2 + w2)
        if order == "row-first":
            for y in range(fhw):
                for x in range(fhw):
                    if grain[i, y // hw2, x // hw2] == 1:
                        fc.append(indices[i, y, x])
                        fp.append(y * fhw + x)
        rows["cc"].append(cc + [content_eos])
        rows["cp"].append(cp + [cpos_eos])
        rows["fc"].append(fc + [content_eos])
        rows["fp"].append(fp + [fpos_eos])

    def pad(lst, value):
        n = max(len(r) for r in lst)
        return np.array([r + [value] * (n - len(r)) for r in lst], dtype=np.int64)

    out = {"coarse_content": pad(rows["cc"], content_pad), "coarse_position": pad(rows["cp"], cpos_pad),
           "fine_content": pad(rows["fc"], content_pad), "fine_position": pad(rows["fp"], fpos_pad)}
    out["coarse_segment"] = np.zeros_like(out["coarse_content"])
    out["fine_segment"] = np.ones_like(out["fine_content"])
    return out


def forward_back(coarse_content, fine_content, coarse_position, fine_position, hw1, hw2, cpos_eos=257, fpos_eos=1025):
    cc, fc = np.asarray(coarse_content, dtype=np.int64), np.asarray(fine_content, dtype=np.int64)
    cp, fp = np.asarray(coarse_position, dtype=np.int64), np.asarray(fine_position, dtype=np.int64)
    b = cc.shape[0]
    fhw = hw1 * hw2
    out = np.zeros((b, fhw * fhw), dtype=np.int64)
    for i in range(b):
        coarse = np.zeros(hw1 * hw1, dtype=np.int64)
        for k in range(cc.shape[1]):
            if cp[i, k] == cpos_eos:
                # the coarse codes reach the map only here (permuter.py:121-124)
                out[i] = coarse.reshape(hw1, hw1).repeat(hw2, axis=0).repeat(hw2, axis=1).reshape(-1)
                break
            coarse[cp[i, k]] = cc[i, k]
        for k in range(fc.shape[1]):
            if fp[i, k] == fpos_eos:
                break
            out[i, fp[i, k]] = fc[i, k]
    return out.reshape(b, fhw, fhw)
