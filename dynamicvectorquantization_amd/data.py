"""Input pipeline with the image transforms on the GPU.

Mirrors /root/reference/data/imagenet_base.py:8-64 (ImagePaths: torchvision transforms on PIL images) and
data/build.py:16-90 (DataModuleFromConfig) for the batch dict {"image": fp32 [B,3,S,S] in [-1,1], "class_label": int64 [B], ...}:

    train:  Resize(256) -> RandomCrop(256) -> RandomHorizontalFlip(0.5) -> ToTensor -> Normalize(0.5, 0.5)
    eval:   Resize(256) -> CenterCrop(256) -> Resize((256, 256)) [a no-op] -> ToTensor -> Normalize(0.5, 0.5)

JPEG decoding runs on host threads (PIL; this image has no rocJPEG); the decoded uint8 pixels of a whole batch go to the device in
ONE pinned copy and two kernels (csrc/imgproc.hip) do the rest -- Pillow's antialiased bilinear resampling reproduced bit for
bit in 8-bit fixed point, computed only for the crop window, with flip / ToTensor / Normalize folded into the second pass.  The
reference's 8 PIL workers manage ~1-2 k images/s per node; at >= 330 images/s per GPU x 8 GPUs that would be the bottleneck.

Resampling coefficients (`resample_coeffs`) follow Pillow's src/libImaging/Resample.c (precompute_coeffs with the bilinear
filter, normalize_coeffs_8bpc with PRECISION_BITS = 22); the target size follows torchvision 0.14's Resize(int): shorter side ->
size, longer side -> int(size * long / short).  RandomCrop / flip decisions come from a numpy Generator (the reference's torch
RNG stream is not reproducible across loaders: parity unpinned by construction; tests inject the decisions).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import queue
import threading

import numpy as np
import torch

from . import _lib
from .kernels import _p, _s, check, lib

PRECISION_BITS = 32 - 8 - 2


# ---- host-side arithmetic of the resize (tiny: O(output size) per image) ---------------------------------------------------
def resized_size(w: int, h: int, size: int):
    """torchvision.transforms.Resize(size:int) on a (w, h) image -> (new_w, new_h) (functional._compute_resized_output_size)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs (bilinear filter, box = whole axis) + normalize_coeffs_8bpc:
    -> (bounds int32 [out_size, 2] = (first input index, count), coeffs int32 [out_size, ksize], ksize)"""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            v = 1.0 - v if v < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            c = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + c * (1 << PRECISION_BITS)) if c < 0 else int(0.5 + c * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_coeff_cache: dict = {}


def _coeffs_cached(in_size, out_size):
    key = (in_size, out_size)
    hit = _coeff_cache.get(key)
    if hit is None:
        if len(_coeff_cache) > 4096:
            _coeff_cache.clear()
        hit = _coeff_cache[key] = resample_coeffs(in_size, out_size)
    return hit


class _Desc(C.Structure):
    """mirror of ImgDesc in csrc/imgproc.hip"""
    _fields_ = [("src_off", C.c_int64), ("tmp_off", C.c_int64), ("w", C.c_int32), ("h", C.c_int32), ("row0", C.c_int32),
                ("rows", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32), ("flip", C.c_int32), ("pad", C.c_int32),
                ("hb_off", C.c_int32), ("hk_off", C.c_int32), ("hks", C.c_int32), ("vb_off", C.c_int32), ("vk_off", C.c_int32),
                ("vks", C.c_int32)]


def plan_batch(images, size, crops=None, flips=None, train=False, rng=None):
    """images: list of uint8 [h,w,3] arrays.  -> dict(src uint8 [bytes], desc uint8 [B * sizeof(ImgDesc)], tab int32 [...],
    tmp_bytes, max_rows): everything the two kernels need, as flat host arrays ready for ONE pinned upload each.
    crops: optional list of (x, y) offsets in the RESIZED image; flips: optional list of bools (else drawn from `rng` when
    train, centre crop / no flip otherwise)."""
    b = len(images)
    descs = (_Desc * b)()
    tabs, tab_len, src_len, tmp_len, max_rows = [], 0, 0, 0, 1
    for i, im in enumerate(images):
        assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, "decoded RGB uint8 [h,w,3] expected"
        h, w = int(im.shape[0]), int(im.shape[1])
        nw, nh = resized_size(w, h, size)
        if crops is not None:
            cx, cy = crops[i]
        elif train:
            cy, cx = int(rng.integers(0, nh - size + 1)), int(rng.integers(0, nw - size + 1))       # RandomCrop.get_params: i, j
        else:
            cy, cx = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))                   # CenterCrop
        flip = bool(flips[i]) if flips is not None else (bool(rng.random() < 0.5) if train else False)
        hb, hk, hks = _coeffs_cached(w, nw)
        vb, vk, vks = _coeffs_cached(h, nh)
        hb, hk = hb[cx:cx + size], hk[cx:cx + size]
        vb, vk = vb[cy:cy + size], vk[cy:cy + size]
        row0 = int(vb[:, 0].min())
        rows = int((vb[:, 0] + vb[:, 1]).max()) - row0
        d = descs[i]
        d.src_off, d.tmp_off, d.w, d.h, d.row0, d.rows = src_len, tmp_len, w, h, row0, rows
        d.crop_x, d.crop_y, d.flip = cx, cy, int(flip)
        d.hb_off, d.hk_off, d.hks = tab_len, tab_len + hb.size, hks
        d.vb_off, d.vk_off, d.vks = tab_len + hb.size + hk.size, tab_len + hb.size + hk.size + vb.size, vks
        tabs += [hb.reshape(-1), hk.reshape(-1), vb.reshape(-1), vk.reshape(-1)]
        tab_len += hb.size + hk.size + vb.size + vk.size
        src_len += h * w * 3
        tmp_len += rows * size * 3
        max_rows = max(max_rows, rows)
    src = np.empty(src_len, dtype=np.uint8)
    for i, im in enumerate(images):
        n = im.shape[0] * im.shape[1] * 3
        src[descs[i].src_off:descs[i].src_off + n] = np.ascontiguousarray(im).reshape(-1)
    return {"src": src, "desc": np.frombuffer(bytes(descs), dtype=np.uint8).copy(), "tab": np.concatenate(tabs).astype(np.int32),
            "tmp_bytes": tmp_len, "max_rows": max_rows, "batch": b, "size": size}


def transform_batch_gpu(plan, device, out=None):
    """run the two kernels on a planned batch -> fp32 [B,3,S,S] on `device` (no synchronisation)"""
    assert C.sizeof(_Desc) == lib().dvq_image_desc_bytes(), "ImgDesc layout mismatch between data.py and imgproc.hip"
    b, s = plan["batch"], plan["size"]

    def up(a):
        t = torch.from_numpy(a)
        return t.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else t.to(device)
    src, desc, tab = up(plan["src"]), up(plan["desc"]), up(plan["tab"])
    tmp = torch.empty(max(1, plan["tmp_bytes"]), dtype=torch.uint8, device=device)
    if out is None:
        out = torch.empty(b, 3, s, s, dtype=torch.float32, device=device)
    check(lib().dvq_image_batch_transform(_p(src), _p(desc), _p(tab), _p(tmp), b, s, plan["max_rows"], _p(out), _s()),
          "dvq_image_batch_transform")
    return out


# ---- datasets / loader (data/imagenet_base.py ImagePaths, data/imagenet.py, data/build.py) ----------------------------------------
def decode_rgb(path) -> np.ndarray:
    """PIL decode -> uint8 [h,w,3] (imagenet_base.py:50-53: non-RGB images are converted)"""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        return np.asarray(im, dtype=np.uint8)


class ImagePaths:
    """file list + labels; `__getitem__` returns the DECODED image and its labels -- the transform happens per batch on the GPU"""

    def __init__(self, paths, labels=None, is_train=False):
        self.paths = list(paths)
        self.labels = dict(labels or {})
        self.labels["file_path_"] = self.paths
        self.is_train = is_train

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        ex = {"image_u8": decode_rgb(self.paths[i])}
        for k, v in self.labels.items():
            ex[k] = v[i]
        return ex


class ImageFolder(ImagePaths):
    """<root>/<class dir>/<image>: class_label = index of the class directory in sorted order (data/imagenet.py:75-82: labels
    are the indices of np.unique(synsets)), relpath, synset"""
    EXTS = (".jpeg", ".jpg", ".png", ".bmp", ".webp")

    def __init__(self, root, is_train=False, limit=None):
        classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        rel, syn = [], []
        for c in classes:
            for f in sorted(os.listdir(os.path.join(root, c))):
                if f.lower().endswith(self.EXTS):
                    rel.append(os.path.join(c, f))
                    syn.append(c)
        if limit:
            rel, syn = rel[:limit], syn[:limit]
        idx = {c: i for i, c in enumerate(sorted(set(syn)))}
        labels = {"relpath": np.array(rel), "synsets": np.array(syn), "class_label": np.array([idx[s] for s in syn], dtype=np.int64)}
        super().__init__([os.path.join(root, r) for r in rel], labels, is_train)


def _imagenet_root(split):
    root = os.environ.get("DVQ_IMAGENET_ROOT")
    if not root:
        raise FileNotFoundError("set DVQ_IMAGENET_ROOT to a directory holding train/ and val/ class folders "
                                "(the reference's data/default.py points at its own cluster path)")
    return os.path.join(root, split)


class ImageNetTrain(ImageFolder):
    """data/imagenet.py:ImageNetTrain(config={is_eval, size}) on $DVQ_IMAGENET_ROOT/train"""

    def __init__(self, config=None):
        self.config = dict(config or {})
        super().__init__(_imagenet_root("train"), is_train=not self.config.get("is_eval", False))


class ImageNetValidation(ImageFolder):
    def __init__(self, config=None):
        self.config = dict(config or {})
        super().__init__(_imagenet_root("val"), is_train=False)


class GpuBatchLoader:
    """iterates a dataset in batches: `num_workers` host threads decode, the transforms of a whole batch run on the GPU, batches
    are produced `prefetch` ahead of the consumer on a side stream"""

    def __init__(self, dataset, batch_size, device, size=256, shuffle=False, num_workers=8, seed=0, drop_last=True, prefetch=2):
        self.ds, self.bs, self.device, self.size = dataset, batch_size, torch.device(device), size
        self.shuffle, self.workers, self.drop_last, self.prefetch = shuffle, max(1, num_workers), drop_last, prefetch
        self.rng = np.random.default_rng(seed)
        self.train = bool(getattr(dataset, "is_train", False))

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _batch(self, idxs, pool):
        ex = list(pool.map(self.ds.__getitem__, idxs))
        plan = plan_batch([e["image_u8"] for e in ex], self.size, train=self.train, rng=self.rng)
        out = {"image": transform_batch_gpu(plan, self.device)}
        for k in ex[0]:
            if k == "image_u8":
                continue
            vals = [e[k] for e in ex]
            out[k] = torch.as_tensor(np.asarray(vals)).to(self.device) if np.asarray(vals).dtype.kind in "iuf" else vals
        return out

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        order = self.rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        batches = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last and batches and len(batches[-1]) < self.bs:
            batches.pop()
        q: queue.Queue = queue.Queue(maxsize=self.prefetch)
        stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

        def producer():
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for idxs in batches:
                        if stream is not None:
                            with torch.cuda.stream(stream):
                                b = self._batch(idxs, pool)
                                ev = torch.cuda.Event()
                                ev.record(stream)
                        else:
                            b, ev = self._batch(idxs, pool), None
                        q.put((b, ev))
                q.put(None)
            except BaseException as e:      # noqa: BLE001 -- surfaced in the consumer thread
                q.put(e)

        threading.Thread(target=producer, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            b, ev = item
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                # the batch was allocated on the producer's stream: tell the caching allocator that the consumer's stream
                # reads it too, so that a dropped batch is not handed back to the producer (and overwritten by the next
                # transform) while the training stream, which runs steps behind the host, still reads it
                for t in b.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
            yield b


class DataModuleFromConfig:
    """data/build.py:16-90: {batch_size, train, validation, test, num_workers} -> *_dataloader() yielding GPU-resident batch dicts"""

    def __init__(self, batch_size, train=None, validation=None, test=None, wrap=False, num_workers=None, train_val=False,
                 device="cuda", size=256):
        from .config import instantiate_from_config
        self.batch_size = batch_size
        self.num_workers = num_workers if num_workers is not None else batch_size * 2
        self.dataset_configs = {k: v for k, v in (("train", train), ("validation", validation), ("test", test)) if v is not None}
        self.device, self.size = device, size
        self.datasets = {k: instantiate_from_config(v) for k, v in self.dataset_configs.items()}
        for k, d in self.datasets.items():
            print("dataset: ", k, len(d))

    def prepare_data(self):
        pass

    def _loader(self, split, shuffle):
        return GpuBatchLoader(self.datasets[split], self.batch_size, self.device, self.size, shuffle=shuffle,
                              num_workers=min(self.num_workers, 32))

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("validation", False)

    def test_dataloader(self):
        return self._loader("test", False)
