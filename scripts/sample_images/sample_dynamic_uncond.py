#!/usr/bin/env python3
"""Unconditional sampling to PNG files -- the reference's scripts/sample_images/sample_dynamic_uncond.py (:21-102): the same sampler
and flags as scripts/sample_val/sample_dynamic_uncond.py, but every sample is written as its own min-max normalised image
`<out>/[fixed_]TopK-..._image/batch_<i>_<j>.png` (torchvision.utils.save_image(..., normalize=True) there) and no pickles.

    python scripts/sample_images/sample_dynamic_uncond.py --yaml_path configs/stage2/uncond_imagenet_p6c18.yml \\
        --model_path last.ckpt --batch_size 50 --sample_num 500 --top_k 300 --top_k_pos 1024
"""
import importlib.util
import os

_twin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sample_val", "sample_dynamic_uncond.py")
_spec = importlib.util.spec_from_file_location("dvq_sample_val_script", _twin)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

if __name__ == "__main__":
    _mod.main(per_image_png=True)
