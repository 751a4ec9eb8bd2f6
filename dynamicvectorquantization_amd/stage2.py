"""Stage-2 (DQ-Transformer) input plumbing on the HIP path: code maps <-> sequences.

Mirrors /root/reference/modules/dynamic_modules/permuter.py:6-135 (DualGrainSeperatePermuter) and
modules/dynamic_modules/label_provider.py:4-92 (SOS providers).  The permutation is integer compaction / scatter work done
by dvq_permute_dual / dvq_permute_dual_back (one workgroup per image); results are bit-exact with the reference, including
its sequential edge semantics (missing EOS, duplicate positions).  StackGPT itself (stackgpt.py) is not on the HIP path yet.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import check
from .kernels import lib


class DualGrainSeperatePermuter(nn.Module):
    def __init__(self, coarse_hw=16, fine_hw=32, content_pad_code=1024, content_eos_code=1025, coarse_position_pad_code=256,
                 coarse_position_eos_code=257, fine_position_pad_code=1024, fine_position_eos_code=1025,
                 fine_position_order="region-first"):
        super().__init__()
        self.hw1, self.hw2, self.fine_hw = coarse_hw, fine_hw // coarse_hw, fine_hw
        self.hw2_square = int(self.hw2 * self.hw2)
        self.content_pad_code, self.content_eos_code = content_pad_code, content_eos_code
        self.coarse_position_pad_code, self.coarse_position_eos_code = coarse_position_pad_code, coarse_position_eos_code
        self.fine_position_pad_code, self.fine_position_eos_code = fine_position_pad_code, fine_position_eos_code
        self.fine_position_order = fine_position_order
        assert self.fine_position_order in ["row-first", "region-first"]

    def forward(self, indices, grain_indices):
        """indices int64 [B,fine_hw,fine_hw], grain_indices int64 [B,coarse_hw,coarse_hw] (0 coarse, 1 fine) -> dict of
        padded sequences (row length = longest sequence of the batch + 1, like pad_sequence)"""
        b = indices.size(0)
        dev = indices.device
        idx = indices.contiguous().long()
        gr = grain_indices.contiguous().long()
        ncell, npix = self.hw1 * self.hw1, self.fine_hw * self.fine_hw
        cc = torch.empty(b, ncell + 1, dtype=torch.long, device=dev)
        cp = torch.empty(b, ncell + 1, dtype=torch.long, device=dev)
        fc = torch.empty(b, npix + 1, dtype=torch.long, device=dev)
        fp = torch.empty(b, npix + 1, dtype=torch.long, device=dev)
        counts = torch.empty(b, 2, dtype=torch.int32, device=dev)
        check(lib().dvq_permute_dual(K._p(idx), K._p(gr), b, self.hw1, self.hw2, 0 if self.fine_position_order == "region-first" else 1,
                                     self.content_pad_code, self.content_eos_code, self.coarse_position_pad_code,
                                     self.coarse_position_eos_code, self.fine_position_pad_code, self.fine_position_eos_code,
                                     K._p(cc), K._p(cp), K._p(fc), K._p(fp), K._p(counts), K._s()), "dvq_permute_dual")
        lc, lf = (counts.max(dim=0)[0] + 1).tolist()          # the one host sync: the ragged batch's row lengths
        cc, cp, fc, fp = cc[:, :lc].contiguous(), cp[:, :lc].contiguous(), fc[:, :lf].contiguous(), fp[:, :lf].contiguous()
        return {"coarse_content": cc, "fine_content": fc, "coarse_position": cp, "fine_position": fp,
                "coarse_segment": torch.zeros_like(cc), "fine_segment": torch.ones_like(fc)}

    def forward_back(self, coarse_content, fine_content, coarse_position, fine_position):
        b, lc = coarse_content.size()
        lf = fine_content.size(1)
        out = torch.empty(b, self.fine_hw, self.fine_hw, dtype=torch.long, device=coarse_content.device)
        args = [t.contiguous().long() for t in (coarse_content, fine_content, coarse_position, fine_position)]
        check(lib().dvq_permute_dual_back(*[K._p(t) for t in args], b, lc, lf, self.hw1, self.hw2, self.coarse_position_eos_code,
                                          self.fine_position_eos_code, K._p(out), K._s()), "dvq_permute_dual_back")
        return out


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class PositionAwareSOSProvider(AbstractEncoder):
    """label_provider.py:11-46: constant [B,1] start tokens"""

    def __init__(self, coarse_sos, coarse_pos_sos, fine_sos=None, fine_pos_sos=None, coarse_seg_sos=None, fine_seg_sos=None):
        super().__init__()
        self.coarse_sos, self.fine_sos = coarse_sos, fine_sos
        self.coarse_pos_sos, self.fine_pos_sos = coarse_pos_sos, fine_pos_sos
        self.activate_seg = coarse_seg_sos is not None
        if self.activate_seg:
            self.coarse_seg_sos, self.fine_seg_sos = coarse_seg_sos, fine_seg_sos

    @staticmethod
    def _const(b, v, device):
        return None if v is None else torch.full((b, 1), int(v), dtype=torch.long, device=device)

    def encode(self, x):
        b, dev = x.size(0), x.device
        out = (self._const(b, self.coarse_sos, dev), self._const(b, self.fine_sos, dev),
               self._const(b, self.coarse_pos_sos, dev), self._const(b, self.fine_pos_sos, dev))
        if self.activate_seg:
            return out + (self._const(b, self.coarse_seg_sos, dev), self._const(b, self.fine_seg_sos, dev))
        return out + (None, None)


class ClassForContentOnlyPositionAwareSOSProvider(AbstractEncoder):
    """label_provider.py:48-92: the content start tokens are the class label + threshold"""

    def __init__(self, n_classes, threshold, coarse_pos_sos, fine_pos_sos=None, coarse_seg_sos=None, fine_seg_sos=None):
        super().__init__()
        self.n_classes, self.threshold = n_classes, threshold
        self.coarse_pos_sos, self.fine_pos_sos = coarse_pos_sos, fine_pos_sos
        self.activate_seg = coarse_seg_sos is not None
        if self.activate_seg:
            self.coarse_seg_sos, self.fine_seg_sos = coarse_seg_sos, fine_seg_sos

    def encode(self, x):
        b, dev = x.size(0), x.device
        c = (x[:, None] + self.threshold).long()
        const = PositionAwareSOSProvider._const
        out = (c, c if self.fine_pos_sos is not None else None, const(b, self.coarse_pos_sos, dev), const(b, self.fine_pos_sos, dev))
        if self.activate_seg:
            return out + (const(b, self.coarse_seg_sos, dev), const(b, self.fine_seg_sos, dev))
        return out + (None, None)


class ClassAwareSOSProvider(AbstractEncoder):
    """label_provider.py:94-128: content and position start tokens are all derived from the class label"""

    def __init__(self, n_classes, threshold_content, threshold_coarse_position, threshold_fine_position, coarse_seg_sos=None,
                 fine_seg_sos=None):
        super().__init__()
        self.n_classes = n_classes
        self.threshold_content, self.threshold_coarse_position = threshold_content, threshold_coarse_position
        self.threshold_fine_position = threshold_fine_position
        self.activate_seg = coarse_seg_sos is not None
        self.coarse_seg_sos, self.fine_seg_sos = coarse_seg_sos, fine_seg_sos

    def encode(self, x):
        b, dev = x.size(0), x.device
        x = x[:, None]
        has_fine = self.fine_seg_sos is not None
        out = (x + self.threshold_content, x + self.threshold_content if has_fine else None,
               x + self.threshold_coarse_position, x + self.threshold_fine_position if has_fine else None)
        if self.activate_seg:
            const = PositionAwareSOSProvider._const
            return out + (const(b, self.coarse_seg_sos, dev), const(b, self.fine_seg_sos, dev))
        return out + (None, None)
