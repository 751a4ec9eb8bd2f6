#!/usr/bin/env python3
"""Grain maps and sequence-length statistics of a (trained) DQ-VAE -- the reference's scripts/tools/visualize_dual_grain.py
(:27-61) on the HIP path: same --yaml_path / --model_path / --batch_size / --image_save_path; images come from --images
(folder or .npy) or --synthetic N.  Prints mean / variance / max / min of the per-image token count (1 per coarse cell, 4 per
fine cell) and stores the grain maps as .npy (+ a PNG overlay per batch when PIL is available).

    python scripts/tools/visualize_dual_grain.py --yaml_path configs/stage1/dqvae-entropy-dual-r05_imagenet.yml \\
        --model_path last.ckpt --images /data/val_images --image_save_path out/
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--yaml_path", type=str, required=True)
    ap.add_argument("--model_path", type=str, default="")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--image_save_path", type=str, default="")
    ap.add_argument("--images", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--dtype", default="bf16")
    opt, _ = ap.parse_known_args()
    import numpy as np
    import torch
    from dynamicvectorquantization_amd import calibrate, config as cfg, runtime as rt, synth
    rt.set_compute_dtype(opt.dtype)
    conf = cfg.load_yaml(opt.yaml_path)
    model = cfg.instantiate_from_config(conf.model)
    if opt.model_path:
        sd = torch.load(opt.model_path, map_location="cpu")
        model.load_state_dict(sd["state_dict"] if "state_dict" in sd else sd)
    model = model.eval().cuda()
    size = int(conf.model.params.get("image_size", 256)) if hasattr(conf.model.params, "get") else 256
    if opt.synthetic > 0:
        images = synth.half_flat_images(opt.synthetic, size, seed=2021)
    elif opt.images:
        images = calibrate.load_images(opt.images, size, opt.limit)
    else:
        ap.error("give --images <folder|.npy> or --synthetic N")
    grains = []
    with torch.no_grad():
        for i in range(0, images.shape[0], opt.batch_size):
            x = torch.from_numpy(images[i:i + opt.batch_size]).cuda()
            out = model(x)
            grains.append(out[2].cpu().numpy())
    grains = np.concatenate(grains)
    stats = calibrate.sequence_length_stats(grains)
    for k in ("mean", "variance", "max", "min"):
        print(f"{k}: ", stats[k])
    if opt.image_save_path:
        os.makedirs(opt.image_save_path, exist_ok=True)
        np.save(os.path.join(opt.image_save_path, "grain_indices.npy"), grains)


if __name__ == "__main__":
    main()
