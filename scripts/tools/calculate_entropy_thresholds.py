#!/usr/bin/env python3
"""Entropy-percentile table for DualGrainFixedEntropyRouter -- the reference's scripts/tools/calculate_entropy_thresholds.py
on the fused HIP entropy kernel.  Same flags and output file; additions: the image source (the reference's LMDB datasets are
out of scope: a folder of images, a .npy, or synthetic half-flat images) and --bins.

    python scripts/tools/calculate_entropy_thresholds.py --dataset_type imagenet --split val --images /data/val_images
    python scripts/tools/calculate_entropy_thresholds.py --synthetic 512 --out /tmp/table.json --bins reference

--bins model (default) calibrates with the bins the model applies at run time (linspace(-1,1,32)), so that
fine_grain_ratito = r yields a fine fraction of r on the calibration set; --bins reference reproduces the reference
script's linspace(0,1,32).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=10)
    ap.add_argument("--dataset_type", type=str, default="ffhq")
    ap.add_argument("--split", type=str, default="val")
    ap.add_argument("--patch_size", type=int, default=16)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--images", type=str, default=None, help="folder of images or .npy [N,3,H,W] in [-1,1]")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic half-flat images instead")
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--bins", choices=["model", "reference"], default="model")
    ap.add_argument("--out", type=str, default=None, help="default: scripts/tools/thresholds/entropy_thresholds_<type>_<split>_patch-<p>.json")
    opt, _ = ap.parse_known_args()
    from dynamicvectorquantization_amd import calibrate, synth
    if opt.synthetic > 0:
        images = synth.half_flat_images(opt.synthetic, opt.image_size, patch=opt.patch_size, seed=2021)
    elif opt.images:
        images = calibrate.load_images(opt.images, opt.image_size, opt.limit)
    else:
        ap.error("give --images <folder|.npy> or --synthetic N")
    ent = calibrate.patch_entropies(images, opt.patch_size, opt.bins, max(1, opt.batch_size))
    print(ent.shape[0])
    table = calibrate.threshold_table(ent)
    out = opt.out or calibrate.default_table_path(opt.dataset_type, opt.split, opt.patch_size)
    calibrate.write_table(out, table)
    print("wrote", out, "median threshold", table["50"])


if __name__ == "__main__":
    main()
