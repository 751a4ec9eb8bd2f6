"""Oracle: the reference's image transforms (numpy).  TEST INFRASTRUCTURE -- never imported by the product package.

Restates what /root/reference/data/imagenet_base.py:16-32 does to a decoded image through torchvision 0.14 / Pillow 9.4
(both pinned in the reference's environment.yml):
  Resize(256): shorter side -> 256, longer side -> int(256 * long / short); PIL `Image.resize(..., BILINEAR)` = antialiased
  separable resampling in 8-bit fixed point (Pillow src/libImaging/Resample.c); Random/CenterCrop(256); optional horizontal
  flip; ToTensor (uint8 / 255 in fp32); Normalize(0.5, 0.5).
`reference_transform` goes through PIL itself (PIL travels to the GPU box, the reference does not); `resample_u8` is the plain
restatement of Pillow's two passes, pinned against PIL in tests/test_data_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resized_size(w, h, size):
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (size, new_long) if w <= h else (new_long, size)


def _coeffs(in_size, out_size):
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) / fscale)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        k = [int(0.5 + (v / ww) * (1 << PRECISION_BITS)) for v in w]
        out.append((xmin, k))
    return out


def _pass(a, coeffs):
    """resample axis 1 of uint8 [rows, n, 3] -> uint8 [rows, len(coeffs), 3]"""
    res = np.empty((a.shape[0], len(coeffs), 3), dtype=np.uint8)
    ai = a.astype(np.int64)
    for xo, (xmin, k) in enumerate(coeffs):
        acc = np.full((a.shape[0], 3), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i, c in enumerate(k):
            acc += ai[:, xmin + i, :] * c
        res[:, xo, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return res


def resample_u8(img, nw, nh):
    """uint8 [h,w,3] -> uint8 [nh,nw,3]: Pillow's horizontal pass, then its vertical pass on the uint8 intermediate"""
    h, w, _ = img.shape
    t = _pass(img, _coeffs(w, nw))                                         # [h, nw, 3]
    return _pass(t.transpose(1, 0, 2), _coeffs(h, nh)).transpose(1, 0, 2)    # [nh, nw, 3]


def finish(resized, size, crop_x, crop_y, flip):
    """crop, flip, ToTensor, Normalize -> fp32 [3, size, size]"""
    a = resized[crop_y:crop_y + size, crop_x:crop_x + size]
    if flip:
        a = a[:, ::-1]
    t = a.astype(np.float32) / np.float32(255.0)
    return ((t - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)


def center_crop_offsets(nw, nh, size):
    return int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))


def reference_transform(img, size=256, crop=None, flip=False):
    """the same through PIL (what torchvision calls): uint8 [h,w,3] -> fp32 [3,size,size]"""
    from PIL import Image
    h, w, _ = img.shape
    nw, nh = resized_size(w, h, size)
    r = np.asarray(Image.fromarray(img, "RGB").resize((nw, nh), Image.BILINEAR), dtype=np.uint8)
    cx, cy = crop if crop is not None else center_crop_offsets(nw, nh, size)
    return finish(r, size, cx, cy, flip)
