"""GPU tests of the entropy-threshold calibration tool: HIP patch entropies for both bin ranges against the
oracle, the resulting percentile tables, the fine ratio the model's router gets from a table calibrated with the model's
bins, and the script end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from dynamicvectorquantization_amd import calibrate, synth
from oracle import entropy as oe

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _images(n=12, size=64):
    rs = np.random.RandomState(3)
    x = synth.half_flat_images(n, size, seed=11)
    x[: n // 3] = np.clip(0.5 * x[: n // 3] + 0.4 * rs.uniform(-1, 1, size=x[: n // 3].shape), -1, 1)     # mid-entropy patches
    return x.astype(np.float32)


@pytest.mark.parametrize("bins", ["model", "reference"])
def test_patch_entropies_and_table_vs_oracle(dev, bins):
    x = _images()
    got = calibrate.patch_entropies(x, 16, bins, batch_size=5)
    ref = oe.patch_entropy(x, 16, bins=calibrate.BINS[bins]).reshape(-1)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)
    tg, tr = calibrate.threshold_table(got), oe.threshold_table(ref)
    assert max(abs(tg[k] - tr[k]) for k in tg) < 1e-4
    if bins == "reference":
        assert not np.allclose(got, calibrate.patch_entropies(x, 16, "model"), atol=1e-3), "the two bin ranges must differ"


def test_calibrated_table_gives_the_nominal_fine_ratio(dev, tmp_path):
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd.dqvae import DualGrainFixedEntropyRouter
    x = _images(16, 64)
    ent = calibrate.patch_entropies(x, 16, "model")
    path = str(tmp_path / "t.json")
    calibrate.write_table(path, calibrate.threshold_table(ent))
    xt = torch.from_numpy(x).to(dev)
    for r in (0.3, 0.5, 0.7):
        router = DualGrainFixedEntropyRouter(path, r)
        h, _ = K.patch_entropy_gate(xt, 16, None)
        gate = router(entropy=h)
        assert abs(float(gate[..., 1].float().mean()) - r) < 0.02, r


def test_script_end_to_end(dev, tmp_path):
    out = str(tmp_path / "table.json")
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts/tools/calculate_entropy_thresholds.py"), "--synthetic", "8",
                        "--image_size", "64", "--batch_size", "3", "--out", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == str(8 * 16)                       # like the reference: prints the number of patches first
    with open(out) as f:
        t = json.load(f)
    assert list(t) == [str(i) for i in range(1, 100)]
    x = synth.half_flat_images(8, 64, patch=16, seed=2021)
    ref = oe.threshold_table(oe.patch_entropy(x, 16).reshape(-1))
    assert max(abs(t[k] - ref[k]) for k in t) < 1e-4


def test_visualize_dual_grain_script(dev, tmp_path):
    """scripts/tools/visualize_dual_grain.py on the shipped entropy-dual YAML (random weights): half-flat images route exactly half
    of the 16 x 16 cells to the fine grain -> 128 + 4 * 128 = 640 tokens per image"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts/tools/visualize_dual_grain.py"), "--yaml_path",
                        "configs/stage1/dqvae-entropy-dual-r05_imagenet.yml", "--synthetic", "4", "--batch_size", "2",
                        "--image_save_path", str(tmp_path)], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = dict(ln.split(":") for ln in r.stdout.strip().splitlines() if ":" in ln)
    assert float(lines["mean"]) == 640.0 and float(lines["variance"]) == 0.0
    assert int(lines["max"]) == 640 and int(lines["min"]) == 640
    g = np.load(os.path.join(str(tmp_path), "grain_indices.npy"))
    assert g.shape == (4, 16, 16) and set(np.unique(g)) == {0, 1}
