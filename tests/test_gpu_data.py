"""GPU parity of the input pipeline: dvq_image_batch_transform vs the reference's transforms run through PIL
(oracle.data.reference_transform = what torchvision 0.14 does on a PIL image) -- BIT-exact fp32 outputs."""
import numpy as np
import pytest
import torch

from dynamicvectorquantization_amd import data as D
from oracle import data as odata
from test_data_cpu import _img

pytestmark = pytest.mark.gpu

SIZES = [(300, 400), (517, 333), (256, 256), (64, 900), (700, 260), (250, 250), (1200, 1600), (257, 256)]


def test_eval_transform_bit_exact(dev):
    imgs = [_img(h, w, 31 * h + w) for h, w in SIZES]
    out = D.transform_batch_gpu(D.plan_batch(imgs, 256), dev).cpu().numpy()
    assert out.shape == (len(imgs), 3, 256, 256) and out.dtype == np.float32
    for i, im in enumerate(imgs):
        ref = odata.reference_transform(im, 256)
        assert np.array_equal(out[i], ref), f"image {i} {im.shape}: {(out[i] != ref).sum()} elements differ"
    assert out.min() >= -1.0 and out.max() <= 1.0


def test_train_transform_bit_exact_with_injected_decisions(dev):
    """RandomCrop offsets and flips injected (the reference draws them from torch's RNG): every window of the resized image,
    mirrored or not, must come out exactly"""
    rs = np.random.RandomState(0)
    imgs = [_img(h, w, 7 * h + w) for h, w in SIZES]
    crops, flips = [], []
    for im in imgs:
        nw, nh = D.resized_size(im.shape[1], im.shape[0], 256)
        crops.append((int(rs.randint(0, nw - 256 + 1)), int(rs.randint(0, nh - 256 + 1))))
        flips.append(bool(rs.randint(0, 2)))
    crops[0], flips[0] = (D.resized_size(400, 300, 256)[0] - 256, 0), True            # far right window, mirrored
    out = D.transform_batch_gpu(D.plan_batch(imgs, 256, crops=crops, flips=flips), dev).cpu().numpy()
    for i, im in enumerate(imgs):
        ref = odata.reference_transform(im, 256, crop=crops[i], flip=flips[i])
        assert np.array_equal(out[i], ref), f"image {i}: {(out[i] != ref).sum()} elements differ"


def test_loader_end_to_end(dev, tmp_path, monkeypatch):
    """DataModuleFromConfig from a `data:` section like the shipped YAMLs': decoded on host threads, transformed on the GPU, batch
    dict {"image", "class_label", ...} resident on the device; eval split bit-exact vs PIL, train split in range with labels"""
    from PIL import Image
    from dynamicvectorquantization_amd import config as cfg
    files = {}
    for split in ("train", "val"):
        for ci, c in enumerate(("n01", "n02", "n03")):
            d = tmp_path / split / c
            d.mkdir(parents=True)
            for j in range(4):
                im = _img(270 + 13 * j, 300 + 29 * ci, ci * 10 + j)
                Image.fromarray(im, "RGB").save(d / f"img{j}.png")
                files[(split, c, j)] = im
    monkeypatch.setenv("DVQ_IMAGENET_ROOT", str(tmp_path))
    dm = cfg.instantiate_from_config({"target": "data.build.DataModuleFromConfig", "params": dict(
        batch_size=4, num_workers=3, device=str(dev),
        train={"target": "data.imagenet.ImageNetTrain", "params": {"config": {"is_eval": False, "size": 256}}},
        validation={"target": "data.imagenet.ImageNetValidation", "params": {"config": {"is_eval": True, "size": 256}}})})
    seen = 0
    for b in dm.val_dataloader():
        assert b["image"].is_cuda and tuple(b["image"].shape) == (4, 3, 256, 256) and b["class_label"].dtype == torch.int64
        for k in range(4):
            c, j = b["synsets"][k], int(b["relpath"][k].split("img")[1].split(".")[0])
            assert np.array_equal(b["image"][k].cpu().numpy(), odata.reference_transform(files[("val", c, j)], 256))
            assert int(b["class_label"][k]) == int(c[2:]) - 1
        seen += 4
    assert seen == 12
    n = 0
    for b in dm.train_dataloader():
        assert float(b["image"].min()) >= -1.0 and float(b["image"].max()) <= 1.0 and b["image"].shape[0] == 4
        n += 1
    assert n == 3


def test_train_py_real_data_checkpoint_resume(dev, tmp_path):
    """`train.py` end to end on image files: the YAML's `data:` section through the GPU input pipeline (--real_data), periodic
    atomic last.ckpt (--save_every), saved project config, and `-r <logdir>` continuing IN that logdir from the saved configs"""
    import os
    import subprocess
    import sys
    import yaml
    from PIL import Image
    from conftest import REPO
    from test_gpu_model import GEOM, model_config
    for split in ("train", "val"):
        for ci, c in enumerate(("n01", "n02")):
            d = tmp_path / "imgs" / split / c
            d.mkdir(parents=True)
            for j in range(6):
                Image.fromarray(_img(70 + 5 * j, 90 + 7 * ci, ci * 10 + j), "RGB").save(d / f"img{j}.png")
    mc = model_config(**GEOM["small"], loss="ae")
    mc["base_learning_rate"] = 1e-5
    cfg_path = tmp_path / "tiny.yml"
    yaml.safe_dump({"model": mc, "data": {"target": "data.build.DataModuleFromConfig", "params": {
        "batch_size": 4, "num_workers": 2,
        "train": {"target": "data.imagenet.ImageNetTrain", "params": {"config": {"is_eval": False, "size": 64}}},
        "validation": {"target": "data.imagenet.ImageNetValidation", "params": {"config": {"is_eval": True, "size": 64}}}}}},
        open(cfg_path, "w"))
    env = dict(os.environ, DVQ_IMAGENET_ROOT=str(tmp_path / "imgs"))
    logs = tmp_path / "logs"
    cmd = [sys.executable, os.path.join(REPO, "train.py"), "-b", str(cfg_path), "--real_data", "--max_epochs", "2", "--max_steps", "5", "--save_every", "2",
           "--logdir", str(logs), "-n", "t"]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    run = [d for d in os.listdir(logs)]
    assert len(run) == 1 and run[0].endswith("_t")
    ck = torch.load(logs / run[0] / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ck["global_step"] == 5 and len(ck["optimizer_states"]) == 1 and "encoder.conv_in.weight" in ck["state_dict"]
    assert os.path.isfile(logs / run[0] / "configs" / "project.yaml")
    assert not os.path.exists(str(logs / run[0] / "checkpoints" / "last.ckpt") + ".tmp")
    # resume: no -b needed (configs come from the logdir), the run continues to step 6 (3 batches per epoch x 2 epochs) in place
    r2 = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "-r", str(logs / run[0]), "--real_data", "--max_epochs", "2"],
                        env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "resumed from" in r2.stdout, r2.stdout[-2000:] + r2.stderr[-3000:]
    ck2 = torch.load(logs / run[0] / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ck2["global_step"] == 6 and len(os.listdir(logs)) == 1
    w1, w2 = ck["state_dict"]["decoder.conv_out.weight"], ck2["state_dict"]["decoder.conv_out.weight"]
    assert not torch.equal(w1, w2)
