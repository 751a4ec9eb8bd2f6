"""CPU tests of the calibration / evaluation helpers: the percentile rule of the reference's
scripts/tools/calculate_entropy_thresholds.py:99-110, the evaluation image transform of data/imagenet_base.py:24-30 and the
sequence-length statistics of scripts/tools/visualize_dual_grain.py:46-56."""
import json
import os

import numpy as np
import pytest

from dynamicvectorquantization_amd import calibrate
from oracle import entropy as oe

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("size", [100, 101, 999, 4096, 65536 + 7])
def test_threshold_table_matches_the_reference_rule(size):
    rs = np.random.RandomState(size)
    ent = rs.uniform(0.0, 3.4, size=size).astype(np.float32)
    got, ref = calibrate.threshold_table(ent), oe.threshold_table(ent)
    assert list(got) == [str(i) for i in range(1, 100)]
    assert got == ref
    srt = np.sort(ent)
    for i in (1, 37, 50, 99):                     # entry "i" = sorted[(size * i) // 100], as a Python float of the fp32 value
        assert got[str(i)] == float(srt[(size * i) // 100])
    vals = [got[str(i)] for i in range(1, 100)]
    assert vals == sorted(vals)


def test_threshold_table_needs_enough_patches():
    with pytest.raises(ValueError):
        calibrate.threshold_table(np.zeros(99, dtype=np.float32))


def test_table_round_trip_and_router_key_arithmetic(tmp_path):
    ent = np.linspace(0.0, 3.0, 1000, dtype=np.float32)
    path = str(tmp_path / "sub" / "table.json")
    calibrate.write_table(path, calibrate.threshold_table(ent))
    with open(path) as f:
        table = json.load(f)
    assert len(table) == 99
    # the router reads key str(int(100 - r * 100)) (RouterDual.py:51): fine fraction r on the calibration set
    for r in (0.3, 0.5, 0.7):
        t = oe.threshold_from_table(path, r)
        assert abs(float((ent > np.float32(t)).mean()) - r) < 0.011
    assert calibrate.default_table_path("imagenet", "train", 16, REPO).endswith(
        "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json")
    assert os.path.exists(calibrate.default_table_path("imagenet", "train", 16, REPO))


def test_shipped_tables_have_the_reference_layout():
    d = os.path.join(REPO, "scripts/tools/thresholds")
    for name in os.listdir(d):
        with open(os.path.join(d, name)) as f:
            t = json.load(f)
        assert list(t) == [str(i) for i in range(1, 100)], name
        v = [t[str(i)] for i in range(1, 100)]
        assert v == sorted(v), name


def test_load_images_npy_and_folder(tmp_path):
    from PIL import Image
    a = np.random.RandomState(0).uniform(-1, 1, size=(3, 3, 32, 32)).astype(np.float32)
    np.save(tmp_path / "x.npy", a)
    assert np.array_equal(calibrate.load_images(str(tmp_path / "x.npy"), 32), a)
    assert calibrate.load_images(str(tmp_path / "x.npy"), 32, limit=2).shape[0] == 2
    # landscape image with a horizontal ramp: Resize(shorter side) + CenterCrop keeps the middle of the ramp
    w, h = 96, 48
    ramp = np.tile(np.linspace(0, 255, w, dtype=np.float32)[None, :, None], (h, 1, 3)).astype(np.uint8)
    os.makedirs(tmp_path / "imgs" / "sub")
    Image.fromarray(ramp).save(tmp_path / "imgs" / "a.png")
    Image.fromarray(ramp[:, :, 0]).save(tmp_path / "imgs" / "sub" / "b_gray.png")          # converted to RGB
    x = calibrate.load_images(str(tmp_path / "imgs"), 24)
    assert x.shape == (2, 3, 24, 24) and x.dtype == np.float32
    assert x.min() >= -1.0 and x.max() <= 1.0
    assert np.allclose(x[0, 0], x[0, 1]) and np.allclose(x[0], x[1], atol=2 / 255)
    mid = x[0, 0, 12]
    assert np.all(np.diff(mid) >= 0) and abs(float(mid.mean())) < 0.05          # centred crop of a symmetric ramp
    assert mid[0] > -0.6 and mid[-1] < 0.6                                      # the outer quarters were cropped away
    with pytest.raises(FileNotFoundError):
        calibrate.load_images(str(tmp_path / "x.npy").replace("x.npy", "nothing_here"), 24)


def test_sequence_length_stats():
    g = np.zeros((3, 4, 4), dtype=np.int64)
    g[1] = 1
    g[2, :2] = 1
    s = calibrate.sequence_length_stats(g)
    assert (s["min"], s["max"]) == (16, 64)
    assert s["mean"] == pytest.approx((16 + 64 + 40) / 3)
    assert s["variance"] == pytest.approx(np.var([16, 64, 40]))
