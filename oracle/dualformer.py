"""Oracle: one teacher-forced stage-2 forward (numpy / torch-CPU).  TEST INFRASTRUCTURE -- never imported by the product package.

Restates /root/reference/models/stage2_dynamic/dqtransformer_uncond_entropy.py:166-215 (`encode_to_z`, `forward`) and the
class-conditional start tokens of modules/dynamic_modules/label_provider.py:94-128 on top of the other oracles: frozen DQ-VAE
(oracle.dqvae) -> code map + grain map -> sequences (oracle.permuter) -> start tokens -> StackGPT losses (oracle.stackgpt).
Pinned against the reference's own `training_step` in tests/test_oracle_golden.py (tests/golden/dualformer.npz).
"""
from __future__ import annotations

import numpy as np
import torch

from . import dqvae as odq
from . import permuter as operm
from . import stackgpt as osg


def start_tokens(kind, b, labels=None, content_sos=514, cpos_sos=18, fpos_sos=66, thr_content=514, thr_cpos=18, thr_fpos=66):
    """-> (c_coarse, c_fine, c_pos_coarse, c_pos_fine, c_seg_coarse, c_seg_fine), each int64 [B,1]"""
    ones = np.ones((b, 1), dtype=np.int64)
    if kind == "uncond":                       # label_provider.py:24-46
        return content_sos * ones, content_sos * ones, cpos_sos * ones, fpos_sos * ones, 0 * ones, 1 * ones
    lab = np.asarray(labels, dtype=np.int64)[:, None]          # label_provider.py:108-128: class label shifted above each vocabulary
    return lab + thr_content, lab + thr_content, lab + thr_cpos, lab + thr_fpos, 0 * ones, 1 * ones


def forward(sd_first, sd_gpt, n_head, x, threshold, kind="uncond", labels=None, hw1=4, hw2=2, order="region-first",
            content_pad=512, content_eos=513, cpos_pad=16, cpos_eos=17, fpos_pad=64, fpos_eos=65):
    """x NCHW fp32 torch tensor -> dict(losses..., z=sequence dict)"""
    with torch.no_grad():
        o = odq.dqvae_forward(sd_first, x, threshold)
    codes = np.asarray(o["codes"]).reshape(x.shape[0], hw1 * hw2, hw1 * hw2)
    grain = o["grain_indices"].numpy()
    z = operm.forward(codes, grain, hw1, hw2, order, content_pad, content_eos, cpos_pad, cpos_eos, fpos_pad, fpos_eos)
    c = start_tokens(kind, x.shape[0], labels)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731
    cc = t(np.concatenate([c[0], z["coarse_content"]], 1))
    fc = t(np.concatenate([c[1], z["fine_content"]], 1))
    cp = t(np.concatenate([c[2], z["coarse_position"]], 1))
    fp = t(np.concatenate([c[3], z["fine_position"]], 1))
    cs = t(np.concatenate([c[4], z["coarse_segment"]], 1))
    fs = t(np.concatenate([c[5], z["fine_segment"]], 1))
    out = osg.forward(sd_gpt, n_head, cc, fc, cp, fp, cs, fs, content_target=torch.cat([cc, fc], 1)[:, 1:],
                      coarse_position_target=cp[:, 1:], fine_position_target=fp, content_pad=content_pad, cpos_pad=cpos_pad,
                      fpos_pad=fpos_pad)
    out["z"] = z
    return out
