"""Feature-routed (Gumbel) grain selection on the HIP kernels.

Mirrors /root/reference/modules/dynamic_modules/RouterDual.py:6-43 (DualGrainFeatureRouter),
RouterTriple.py:6-56 (TripleGrainFeatureRouter) and the routing tail of EncoderDual.py:130-156 /
EncoderTriple.py:143-183 (gumbel_softmax(hard) -> argmax -> nearest-upsampled select -> gate_grad scaling ->
codebook mask).

Heads are NHWC tensors ordered coarsest -> finest.  The per-cell routing logits ([B,hc,wc,S], a few thousand
numbers) go through the Gumbel / straight-through arithmetic as host-side torch ops with a LOCAL autograd graph (the same
treatment as the PatchGAN logit maps in losses.py); everything that touches feature maps is a kernel:
GroupNorm (dvq_gn_*), pooling into the concatenated router row (dvq_avgpool_slice), the MLP (dvq_gemm_* + dvq_silu),
the merge and its backward (dvq_grain_merge*).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from .layers import Linear, Normalize, Tape, _child


class _FeatureRouter(nn.Module):
    """shared implementation; subclasses fix the head names / parameter names of the reference"""

    HEADS = ()          # finest-last names, e.g. ("coarse", "fine")

    def __init__(self, num_channels, normalization_type="none", gate_type="1layer-fc", relu_ok=False):
        super().__init__()
        s = len(self.HEADS)
        self.num_splits, self.num_channels, self.gate_type = s, num_channels, gate_type
        if gate_type == "1layer-fc":
            self.gate = Linear(num_channels * s, s)
        elif gate_type == "2layer-fc-SiLu":
            self.gate = nn.Sequential(Linear(num_channels * s, num_channels * s), nn.SiLU(inplace=True), Linear(num_channels * s, s))
        elif gate_type == "2layer-fc-ReLu" and relu_ok:          # RouterTriple.py:23-28 (the dual router has no such branch)
            self.gate = nn.Sequential(Linear(num_channels * s, num_channels * s), nn.ReLU(inplace=True), Linear(num_channels * s, s))
        else:
            raise NotImplementedError()
        self.normalization_type = normalization_type
        for name in reversed(self.HEADS):          # registration (= state_dict) order of the reference: fine first
            if normalization_type == "none":
                setattr(self, f"feature_norm_{name}", nn.Identity())
            elif "group" in normalization_type:
                groups = int(normalization_type.split("-")[-1])
                setattr(self, f"feature_norm_{name}", Normalize(num_channels, num_groups=groups, eps=1e-6))
            else:
                raise NotImplementedError()

    # heads: [coarsest .. finest] NHWC -> logits fp32 [B,hc,wc,S]
    def fwd(self, heads, tape):
        s, c = self.num_splits, self.num_channels
        b, hc, wc, _ = heads[0].shape
        feat = torch.empty(b, hc, wc, s * c, dtype=heads[0].dtype, device=heads[0].device)
        for lvl, (name, h) in enumerate(zip(self.HEADS, heads)):
            norm = getattr(self, f"feature_norm_{name}")
            if isinstance(norm, Normalize):
                h = norm.fwd(h, _child(tape, f"n{lvl}"), silu=False)
            # torch.cat([h_coarse, avg(h_median), avg(h_fine)], dim=1): level l sits in channel slice l
            K.avgpool_slice(h, 1 << lvl, feat, lvl * c)
        x = feat.view(b * hc * wc, s * c)
        if isinstance(self.gate, Linear):
            y = self.gate.fwd(x, _child(tape, "g0"))
        else:
            hid = self.gate[0].fwd(x, _child(tape, "g0"))
            act = K.relu(hid) if isinstance(self.gate[1], nn.ReLU) else K.silu(hid)
            if tape is not None:
                tape.s["hid"] = hid
            y = self.gate[2].fwd(act, _child(tape, "g2"))
        if tape is not None:
            tape.s["shape"] = (b, hc, wc)
        return y[:, :s].float().reshape(b, hc, wc, s)

    def bwd(self, dlogits, tape):
        """dlogits fp32 [B,hc,wc,S] -> [d head_l]"""
        s, c = self.num_splits, self.num_channels
        b, hc, wc = tape.s["shape"]
        last = self.gate if isinstance(self.gate, Linear) else self.gate[2]
        cd = tape.child("g0").s["x"].dtype
        dy = torch.zeros(b * hc * wc, last.out_p, dtype=cd, device=dlogits.device)
        dy[:, :s] = dlogits.reshape(-1, s).to(cd)
        if isinstance(self.gate, Linear):
            dfeat = self.gate.bwd(dy, tape.child("g0"))
        else:
            dact = self.gate[2].bwd(dy, tape.child("g2"))
            dhid = K.relu_bwd(tape.s["hid"], dact) if isinstance(self.gate[1], nn.ReLU) else K.silu_bwd(tape.s["hid"], dact)
            dfeat = self.gate[0].bwd(dhid, tape.child("g0"))
        dfeat = dfeat.view(b, hc, wc, s * c)
        out = []
        for lvl, name in enumerate(self.HEADS):
            g = K.avgpool_slice_bwd(dfeat, lvl * c, c, 1 << lvl)
            norm = getattr(self, f"feature_norm_{name}")
            if isinstance(norm, Normalize):
                g = norm.bwd(g, tape.child(f"n{lvl}"))
            out.append(g)
        return out


class DualGrainFeatureRouter(_FeatureRouter):
    """RouterDual.py:6-43: forward(h_fine, h_coarse) -> logits [B,hc,wc,2] (index 0 = coarse, 1 = fine)"""
    HEADS = ("coarse", "fine")

    def __init__(self, num_channels, normalization_type="none", gate_type="1layer-fc"):
        super().__init__(num_channels, normalization_type, gate_type)


class TripleGrainFeatureRouter(_FeatureRouter):
    """RouterTriple.py:6-56: logits [B,hc,wc,3] (0 coarse, 1 median, 2 fine)"""
    HEADS = ("coarse", "median", "fine")

    def __init__(self, num_channels, normalization_type="none", gate_type="1layer-fc"):
        super().__init__(num_channels, normalization_type, gate_type, relu_ok=True)


# ---------------------------------------------------------------------------------------------
def gumbel_softmax_hard(logits, exponential=None, tau=1.0):
    """torch.nn.functional.gumbel_softmax(logits, tau, hard=True, dim=-1) restated so that the noise can be injected:
    gumbels = -log(E), E ~ Exp(1); y_soft = softmax((logits + gumbels) / tau); ret = y_hard - sg(y_soft) + y_soft"""
    e = torch.empty_like(logits).exponential_() if exponential is None else exponential.to(logits)
    y_soft = ((logits - e.log()) / tau).softmax(dim=-1)
    index = y_soft.max(dim=-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
    return y_hard - y_soft.detach() + y_soft


class Routing:
    """State of one routed forward: the local autograd graph over the logits and what the merge needs."""

    def __init__(self, logits, stochastic, want_grad, exponential=None):
        with torch.enable_grad():
            self.leaf = logits.detach().requires_grad_(bool(want_grad))
            gate = gumbel_softmax_hard(self.leaf, exponential) if stochastic else self.leaf
            self.gate = gate.permute(0, 3, 1, 2)                                   # [B,S,hc,wc] (reference layout)
            self.gate_grad = self.gate.max(dim=1)[0] if stochastic else None       # [B,hc,wc]: value 1, carries d/d y_soft
        self.indices = self.gate.detach().argmax(dim=1).contiguous()               # int64 [B,hc,wc]
        self.scale = self.gate_grad.detach().float().contiguous() if stochastic else None

    def backward(self, g_gate, d_gate_grad):
        """-> d logits (fp32 [B,hc,wc,S]) from the gradient w.r.t. the returned gate and w.r.t. gate_grad"""
        outs, grads = [], []
        if g_gate is not None:
            outs.append(self.gate)
            grads.append(g_gate.to(self.gate.dtype))
        if d_gate_grad is not None and self.gate_grad is not None:
            outs.append(self.gate_grad)
            grads.append(d_gate_grad.to(self.gate_grad.dtype))
        if not outs or not self.leaf.requires_grad:
            return torch.zeros_like(self.leaf)
        (g,) = torch.autograd.grad(outs, [self.leaf], grads, allow_unused=True)
        return torch.zeros_like(self.leaf) if g is None else g
