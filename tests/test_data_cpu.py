"""CPU: the input pipeline's host logic (resize geometry, Pillow coefficient tables, batch planning) and the oracle's restatement
of Pillow's resampling, pinned against PIL itself."""
import numpy as np
import pytest

from dynamicvectorquantization_amd import data as D
from oracle import data as odata


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(h // 7 + 2, w // 7 + 2, 3)).astype(np.uint8)
    big = np.kron(base, np.ones((7, 7, 1), dtype=np.uint8))[:h, :w]
    return np.clip(big.astype(np.int32) + rs.randint(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", [(300, 400), (517, 333), (256, 256), (64, 900), (700, 260), (250, 250)])
def test_oracle_resample_equals_pil(h, w):
    from PIL import Image
    img = _img(h, w, h * 1000 + w)
    nw, nh = odata.resized_size(w, h, 256)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((nw, nh), Image.BILINEAR), dtype=np.uint8)
    assert ref.shape == (nh, nw, 3)
    assert np.array_equal(odata.resample_u8(img, nw, nh), ref)            # bit-exact, up- and down-scaling


def test_resized_size_is_torchvisions():
    # torchvision 0.14 Resize(int): long side = int(size * long / short) -- truncation, not rounding
    assert D.resized_size(500, 375, 256) == (341, 256) and D.resized_size(375, 500, 256) == (256, 341)
    assert D.resized_size(1000, 333, 256) == (768, 256) and D.resized_size(256, 256, 256) == (256, 256)
    assert D.resized_size(333, 1001, 256) == (256, int(256 * 1001 / 333))
    for w, h in ((500, 375), (123, 457), (256, 999)):
        assert D.resized_size(w, h, 256) == odata.resized_size(w, h, 256)


@pytest.mark.parametrize("n_in,n_out", [(400, 341), (333, 256), (256, 256), (100, 256), (2000, 256)])
def test_coefficient_tables_match_the_oracle(n_in, n_out):
    bounds, kk, ksize = D.resample_coeffs(n_in, n_out)
    ref = odata._coeffs(n_in, n_out)
    assert bounds.shape == (n_out, 2) and kk.shape == (n_out, ksize)
    for xx, (xmin, k) in enumerate(ref):
        assert bounds[xx, 0] == xmin and bounds[xx, 1] == len(k) and list(kk[xx, :len(k)]) == k and not kk[xx, len(k):].any()
        assert abs(int(kk[xx].sum()) - (1 << 22)) <= len(k)               # normalised weights in 22-bit fixed point


def test_plan_batch_layout():
    imgs = [_img(300, 400, 1), _img(517, 333, 2), _img(256, 256, 3)]
    plan = D.plan_batch(imgs, 256, crops=[(10, 0), (0, 31), (0, 0)], flips=[False, True, False])
    assert plan["batch"] == 3 and plan["src"].size == sum(i.size for i in imgs)
    import ctypes as C
    descs = (D._Desc * 3).from_buffer_copy(plan["desc"].tobytes())
    assert descs[1].src_off == imgs[0].size and (descs[1].w, descs[1].h, descs[1].flip) == (333, 517, 1)
    assert descs[0].crop_x == 10 and descs[1].crop_y == 31
    # the vertical pass of image 1 reads rows [row0, row0 + rows) only: the crop starts 31 resized rows down
    assert descs[1].row0 > 0 and descs[1].rows < 517 and plan["max_rows"] == max(d.rows for d in descs)
    assert plan["tmp_bytes"] == sum(d.rows for d in descs) * 256 * 3
    # centre crop / no flip defaults (eval transform), RNG decisions when training
    ev = D.plan_batch(imgs[:1], 256)
    d0 = (D._Desc * 1).from_buffer_copy(ev["desc"].tobytes())[0]
    assert (d0.crop_x, d0.crop_y, d0.flip) == (odata.center_crop_offsets(341, 256, 256) + (0,))
    tr = D.plan_batch(imgs, 256, train=True, rng=np.random.default_rng(0))
    dt = (D._Desc * 3).from_buffer_copy(tr["desc"].tobytes())
    assert all(0 <= d.crop_x <= D.resized_size(d.w, d.h, 256)[0] - 256 for d in dt)


def test_image_folder_and_plugin_targets(tmp_path, monkeypatch):
    from PIL import Image
    from dynamicvectorquantization_amd import config as cfg
    for split in ("train", "val"):
        for ci, c in enumerate(("n02", "n01")):
            d = tmp_path / split / c
            d.mkdir(parents=True)
            for j in range(3):
                Image.fromarray(_img(40 + 10 * j, 50, ci * 10 + j), "RGB").save(d / f"img{j}.png")
    monkeypatch.setenv("DVQ_IMAGENET_ROOT", str(tmp_path))
    ds = cfg.instantiate_from_config({"target": "data.imagenet.ImageNetTrain", "params": {"config": {"is_eval": False, "size": 256}}})
    assert len(ds) == 6 and ds.is_train and list(ds.labels["class_label"]) == [0, 0, 0, 1, 1, 1]      # sorted synsets: n01 < n02
    ex = ds[4]
    assert ex["image_u8"].dtype == np.uint8 and ex["image_u8"].shape == (50, 50, 3) and ex["synsets"] == "n02" and ex["class_label"] == 1
    va = cfg.instantiate_from_config({"target": "data.imagenet.ImageNetValidation", "params": {"config": {"is_eval": True, "size": 256}}})
    assert len(va) == 6 and not va.is_train
    assert cfg.get_obj_from_str("data.build.DataModuleFromConfig") is D.DataModuleFromConfig
