"""Host-side mirror of the DQ-VAE model family on libdvq_hip kernels.

Mirrors (same class names, constructor kwargs, parameter names/shapes, call signatures):
  * Entropy, DualGrainVQModel      /root/reference/models/stage1_dynamic/dqvae_dual_entropy.py:13-262
  * DualGrainFixedEntropyRouter    /root/reference/modules/dynamic_modules/RouterDual.py:46-57
  * DualGrainEncoder               /root/reference/modules/dynamic_modules/EncoderDual.py:16-156
  * Decoder, PositionEmbedding2DLearned  /root/reference/modules/dynamic_modules/DecoderPositional.py:13-146
  * FourierPositionEmbedding       /root/reference/modules/dynamic_modules/fourier_embedding.py:7-55

Data layout: images stay NCHW fp32 at the API (the reference's batch format); inside the model every
activation is NHWC in the runtime compute dtype, 3-channel images are zero-padded to one 16-byte
channel vector.  The whole autoencoder runs as ONE autograd node (`_AEFn`); its backward is an explicit
reverse walk over the per-layer tapes.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import kernels as K
from . import runtime as rt
from .config import instantiate_from_config
from .layers import (AttnBlock, Conv2d, Downsample, HipModule, Normalize, ResnetBlock, Tape, Upsample, _child,
                     _grad_buf, norm_swish_conv, to_nchw, to_nhwc)
from .routing import DualGrainFeatureRouter, Routing, TripleGrainFeatureRouter, _FeatureRouter  # noqa: F401


# ---------------------------------------------------------------------------------------------
class Entropy(nn.Sequential):
    """dqvae_dual_entropy.py:13-63: per-patch soft-histogram entropy.  forward(x NCHW fp32) -> [B,h,w]."""

    def __init__(self, patch_size, image_width, image_height):
        super().__init__()
        self.width, self.height, self.psize = image_width, image_height, patch_size
        self.patch_num = int(self.width * self.height / self.psize ** 2)
        self.hw = int(self.width // self.psize)

    def forward(self, inputs):
        x = inputs.contiguous().float() if inputs.dtype != torch.float32 or not inputs.is_contiguous() else inputs
        assert x.shape[2] == self.height and x.shape[3] == self.width, (x.shape, self.height, self.width)
        ent, _ = K.patch_entropy_gate(x, self.psize, None)
        return ent


class DualGrainFixedEntropyRouter(nn.Module):
    """RouterDual.py:46-57 (incl. the reference's key arithmetic int(100 - r*100) and its spelling)."""

    def __init__(self, json_path, fine_grain_ratito):
        super().__init__()
        if not os.path.isabs(json_path) and not os.path.exists(json_path):
            # the shipped YAMLs name the table relative to the repository root (the reference is run from there): resolve it against
            # this checkout when the process was started elsewhere
            alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), json_path)
            json_path = alt if os.path.exists(alt) else json_path
        with open(json_path, "r", encoding="utf-8") as f:
            content = json.load(f)
        self.fine_grain_threshold = content["{}".format(str(int(100 - fine_grain_ratito * 100)))]

    def forward(self, h_fine=None, h_coarse=None, entropy=None):
        """entropy [B,h,w] fp32 -> gate int64 [B,h,w,2] = [coarse, fine]"""
        t = float(self.fine_grain_threshold)      # a Python scalar: no host->device copy (= no sync) per forward
        return torch.stack([entropy <= t, entropy > t], dim=-1).long()


# ---------------------------------------------------------------------------------------------
class _GrainEncoder(HipModule):
    """Shared CNN trunk + S output heads + routing of DualGrainEncoder (EncoderDual.py:16-156) and TripleGrainEncoder
    (EncoderTriple.py:14-183).  HEADS lists the head names coarsest -> finest; parameter names follow the reference
    (`mid_coarse.block_1...`, `norm_out_fine`, `conv_out_median`, ...)."""

    HEADS = ("coarse", "fine")
    OUT_KEY = "h_dual"

    def __init__(self, *, ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resamp_with_conv, in_channels,
                 resolution, z_channels, router_config, update_router):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, 3, 1, 1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        # heads, coarsest first; the head k levels above the bottom has block_in / (ch_mult[-k] // ch_mult[-k-1]) channels
        width = block_in
        for k, name in enumerate(self.HEADS):
            if k > 0:
                width = width // (ch_mult[-k] // ch_mult[-k - 1])
            mid = nn.Module()
            mid.block_1 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
            mid.attn_1 = AttnBlock(width)
            mid.block_2 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
            setattr(self, f"mid_{name}", mid)
            setattr(self, f"norm_out_{name}", Normalize(width))
            setattr(self, f"conv_out_{name}", Conv2d(width, z_channels, 3, 1, 1))
        self.router = instantiate_from_config(router_config)
        self._grad_hook = None             # set by the Trainer under data parallelism: called with parameters whose gradients are final
        self.update_router = update_router
        self.feature_routed = isinstance(self.router, _FeatureRouter)
        self.gumbel_exponential = None     # test hook: Exp(1) noise [B,hc,wc,S] replacing the device RNG draw

    # NHWC core -------------------------------------------------------------------------------------
    def fwd(self, x_img, grain, tape):
        """x_img: NCHW fp32 image; grain: int64 [B,hc,wc] level map for the fixed-entropy router (None when feature
        routed).  Returns (merged NHWC, codebook mask [B,hf,wf], Routing or None)."""
        cd = rt.compute_dtype()
        s = len(self.HEADS)
        x = K.nchw_to_nhwc_pad(x_img, K.vec(cd) * -(-self.in_channels // K.vec(cd)), cd)
        h = self.conv_in.fwd(x, _child(tape, "conv_in"))
        taps = {}
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = lvl.block[i_block].fwd(h, _child(tape, f"d{i_level}b{i_block}"))
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].fwd(h, _child(tape, f"d{i_level}a{i_block}"))
            k = self.num_resolutions - 1 - i_level           # head k taps the trunk k levels above the bottom
            if 0 < k < s:
                taps[k] = h
            if i_level != self.num_resolutions - 1:
                h = lvl.downsample.fwd(h, _child(tape, f"d{i_level}ds"))
        taps[0] = h
        heads = []
        for k, name in enumerate(self.HEADS):
            mid = getattr(self, f"mid_{name}")
            t = mid.block_1.fwd(taps[k], _child(tape, f"m{k}1"))
            t = mid.attn_1.fwd(t, _child(tape, f"m{k}a"))
            t = mid.block_2.fwd(t, _child(tape, f"m{k}2"))
            heads.append(norm_swish_conv(getattr(self, f"norm_out_{name}"), getattr(self, f"conv_out_{name}"), t, tape,
                                         f"no{k}", f"co{k}"))
        routing = None
        if self.feature_routed:
            logits = self.router.fwd(heads, _child(tape, "router"))
            routing = Routing(logits, stochastic=self.training and self.update_router, want_grad=tape is not None,
                              exponential=self.gumbel_exponential)
            merged, mask = K.grain_merge(heads, routing.indices, routing.scale)
            if tape is not None:
                tape.s.update(routing=routing, heads=heads)
        else:
            assert s == 2, "fixed-entropy routing is dual grain"
            merged, mask = K.dual_merge(heads[1], heads[0], grain)
            if tape is not None:
                tape.s["grain"] = grain
        return merged, mask, routing

    def bwd(self, g_merged, tape, g_gate=None):
        s = len(self.HEADS)
        if self.feature_routed:
            routing, heads = tape.s["routing"], tape.s["heads"]
            gh, dscale = K.grain_merge_bwd(g_merged, heads, routing.indices, routing.scale, want_dscale=routing.scale is not None)
            dlogits = routing.backward(g_gate, dscale)
            gr = self.router.bwd(dlogits, tape.child("router"))
            gh = [K.add(a, b) for a, b in zip(gh, gr)]
        else:
            gf, gc = K.dual_merge_bwd(g_merged, tape.s["grain"])
            gh = [gc, gf]
        gt = {}
        for k, name in enumerate(self.HEADS):
            mid = getattr(self, f"mid_{name}")
            g = getattr(self, f"conv_out_{name}").bwd(gh[k], tape.child(f"co{k}"))
            g = getattr(self, f"norm_out_{name}").bwd(g, tape.child(f"no{k}"))
            g = mid.block_2.bwd(g, tape.child(f"m{k}2"))
            g = mid.attn_1.bwd(g, tape.child(f"m{k}a"))
            gt[k] = mid.block_1.bwd(g, tape.child(f"m{k}1"))
        hook = self._grad_hook
        if hook is not None:               # heads + router are done: their gradient exchange overlaps the trunk's backward
            done = list(self.router.parameters())
            for name in self.HEADS:
                for mod in (getattr(self, f"mid_{name}"), getattr(self, f"norm_out_{name}"), getattr(self, f"conv_out_{name}")):
                    done += list(mod.parameters())
            hook(done)
        g = gt[0]
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.down[i_level]
            if i_level != self.num_resolutions - 1:
                g = lvl.downsample.bwd(g, tape.child(f"d{i_level}ds"))
            k = self.num_resolutions - 1 - i_level
            if 0 < k < s:
                g = K.add(g, gt[k])
            for i_block in reversed(range(self.num_res_blocks)):
                if len(lvl.attn) > 0:
                    g = lvl.attn[i_block].bwd(g, tape.child(f"d{i_level}a{i_block}"))
                g = lvl.block[i_block].bwd(g, tape.child(f"d{i_level}b{i_block}"))
            if hook is not None:
                hook(list(lvl.parameters()))
        self.conv_in.bwd(g, tape.child("conv_in"), need_dx=False)
        return None

    # reference signature ---------------------------------------------------------------------------
    def forward(self, x, x_entropy=None):
        assert x.shape[2] == x.shape[3] == self.resolution, "{}, {}, {}".format(x.shape[2], x.shape[3], self.resolution)
        params = [p for p in self.parameters() if p.requires_grad]
        if self.feature_routed:
            h, mask, gate, indices = _EncFn.apply(self, torch.is_grad_enabled(), x, None, *params)
        else:
            gate = self.router(h_fine=None, h_coarse=None, entropy=x_entropy).permute(0, 3, 1, 2)
            indices = gate.argmax(dim=1)
            h, mask, _, _ = _EncFn.apply(self, torch.is_grad_enabled(), x, indices.contiguous(), *params)
        return {self.OUT_KEY: h, "indices": indices, "codebook_mask": mask, "gate": gate}


class DualGrainEncoder(_GrainEncoder):
    """EncoderDual.py:16-156 (fixed-entropy router, or feature router with Gumbel straight-through when
    update_router=True and training)."""
    HEADS = ("coarse", "fine")
    OUT_KEY = "h_dual"

    def __init__(self, *, ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, router_config=None, update_router=True,
                 **ignore_kwargs):
        super().__init__(ch=ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks, attn_resolutions=attn_resolutions,
                         dropout=dropout, resamp_with_conv=resamp_with_conv, in_channels=in_channels, resolution=resolution,
                         z_channels=z_channels, router_config=router_config, update_router=update_router)


class TripleGrainEncoder(_GrainEncoder):
    """EncoderTriple.py:14-183 (always Gumbel-routed in training)."""
    HEADS = ("coarse", "median", "fine")
    OUT_KEY = "h_triple"

    def __init__(self, *, ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, router_config=None, **ignore_kwargs):
        super().__init__(ch=ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks, attn_resolutions=attn_resolutions,
                         dropout=dropout, resamp_with_conv=resamp_with_conv, in_channels=in_channels, resolution=resolution,
                         z_channels=z_channels, router_config=router_config, update_router=True)


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, want_grad, x, grain, *params):
        ctx.module, ctx.tape, ctx.n = module, (Tape() if want_grad else None), len(params)
        with torch.no_grad():
            merged, mask, routing = module.fwd(x.contiguous().float(), grain, ctx.tape)
            out = to_nchw(K.cast(merged, torch.float32))
        mask = mask.unsqueeze(1)
        if routing is None:
            ctx.mark_non_differentiable(mask)
            return out, mask, None, None
        gate, idx = routing.gate.detach(), routing.indices
        ctx.mark_non_differentiable(mask, idx)
        return out, mask, gate, idx

    @staticmethod
    def backward(ctx, g, _gm, g_gate=None, _gi=None):
        with torch.no_grad():
            ctx.module.bwd(to_nhwc(g, rt.compute_dtype()), ctx.tape, g_gate)
        return (None, None, None, None) + (None,) * ctx.n


# ---------------------------------------------------------------------------------------------
def convert_to_coord_format(b, h, w, device="cpu", integer_values=False):
    if integer_values:
        x_channel = torch.arange(w, dtype=torch.float, device=device).view(1, 1, 1, -1).repeat(b, 1, w, 1)
        y_channel = torch.arange(h, dtype=torch.float, device=device).view(1, 1, -1, 1).repeat(b, 1, 1, h)
    else:
        x_channel = torch.linspace(-1, 1, w, device=device).view(1, 1, 1, -1).repeat(b, 1, w, 1)
        y_channel = torch.linspace(-1, 1, h, device=device).view(1, 1, -1, 1).repeat(b, 1, 1, h)
    return torch.cat((x_channel, y_channel), dim=1)


class _ParamConv1x1(nn.Module):
    """parameter holder with nn.Conv2d names (`weight` [Cout,Cin,1,1], `bias`) for tiny host-side maps"""

    def __init__(self, ch_in, ch_out, bound):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(ch_out, ch_in, 1, 1).uniform_(-bound, bound))
        b = 1 / math.sqrt(ch_in)
        self.bias = nn.Parameter(torch.empty(ch_out).uniform_(-b, b))


class ConLinear(nn.Module):
    def __init__(self, ch_in, ch_out, is_first=False, bias=True):
        super().__init__()
        self.conv = _ParamConv1x1(ch_in, ch_out, np.sqrt(9 / ch_in) if is_first else np.sqrt(3 / ch_in))


class LFF(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.ffm = ConLinear(2, hidden_size, is_first=True)


class FourierPositionEmbedding(nn.Module):
    """fourier_embedding.py:45-55: x + sin(Conv1x1([x;y] coords)).  The [C,h,w] bias does not depend on
    the batch, so it is evaluated once per step on the 2xC parameters (1024 x 2 x 256 MACs: host-side
    torch, not a kernel) and added by dvq_add_bias_bcast; its gradient is a batch reduction
    (dvq_sum_batch) followed by the same tiny map."""

    def __init__(self, coord_size, hidden_size, integer_values=False):
        super().__init__()
        # constant grid: a non-persistent buffer so that it moves with the module (an H2D copy per forward would be a
        # pageable-memory copy, i.e. a host<->device synchronisation in the middle of every step)
        self.register_buffer("coord", convert_to_coord_format(1, coord_size, coord_size, "cpu", integer_values), persistent=False)
        self.lff = LFF(hidden_size)

    def bias_hwc(self, device):
        if self.coord.device != torch.device(device):
            self.coord = self.coord.to(device)
        coord = self.coord[0]                                                # [2,h,w]
        w = self.lff.ffm.conv.weight[:, :, 0, 0]                             # [C,2]
        # 2 input channels: two broadcast multiply-adds (no GEMM library call)
        pre = (coord[0].unsqueeze(-1) * w[:, 0] + coord[1].unsqueeze(-1) * w[:, 1]) + self.lff.ffm.conv.bias
        return torch.sin(pre), pre, coord

    def accumulate_grad(self, g_hwc, pre, coord):
        gp = g_hwc * torch.cos(pre)                                           # [h,w,C]
        gw = torch.stack([(gp * coord[0].unsqueeze(-1)).sum(dim=(0, 1)), (gp * coord[1].unsqueeze(-1)).sum(dim=(0, 1))], dim=1)
        _grad_buf(self.lff.ffm.conv.weight).add_(gw[:, :, None, None])
        _grad_buf(self.lff.ffm.conv.bias).add_(gp.sum(dim=(0, 1)))


class PositionEmbedding2DLearned(nn.Module):
    """DecoderPositional.py:13-39 (trunc-normal init, tools.py:40-57)."""

    def __init__(self, n_row, feats_dim, n_col=None):
        super().__init__()
        n_col = n_col if n_col is not None else n_row
        self.row_embed = nn.Embedding(n_row, feats_dim)
        self.col_embed = nn.Embedding(n_col, feats_dim)
        nn.init.trunc_normal_(self.row_embed.weight, mean=0.0, std=1.0, a=-2.0, b=2.0)
        nn.init.trunc_normal_(self.col_embed.weight, mean=0.0, std=1.0, a=-2.0, b=2.0)

    def bias_hwc(self, h, w):
        return self.col_embed.weight[:w].unsqueeze(0) + self.row_embed.weight[:h].unsqueeze(1)   # [h,w,C]

    def accumulate_grad(self, g_hwc):
        h, w, _ = g_hwc.shape
        _grad_buf(self.col_embed.weight)[:w].add_(g_hwc.sum(dim=0))
        _grad_buf(self.row_embed.weight)[:h].add_(g_hwc.sum(dim=1))


class Decoder(HipModule):
    """DecoderPositional.py:41-146 (position_type 'fourier+learned', 'fourier', 'learned')."""

    def __init__(self, ch, in_ch, out_ch, ch_mult, num_res_blocks, resolution, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, give_pre_end=False, latent_size=32, window_size=2, position_type="relative"):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_ch, self.ch = num_res_blocks, resolution, in_ch, ch
        self.temb_ch, self.give_pre_end, self.out_ch = 0, give_pre_end, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, in_ch, curr_res, curr_res)
        self.conv_in = Conv2d(in_ch, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, 3, 1, 1)
        self.position_type = position_type
        # "learned" and "learned-relative": the reference registers the parameter and its forward has no branch that uses it
        # (DecoderPositional.py:94-99 against :112-123) -- the same here: present in the state_dict, nothing added, no gradient
        if position_type == "learned":
            self.position_bias = PositionEmbedding2DLearned(n_row=latent_size, feats_dim=in_ch)
        elif position_type == "learned-relative":
            self.position_bias = PositionEmbedding2DLearned(n_row=window_size, feats_dim=in_ch)
            self.window_size, self.window_num = window_size, latent_size // window_size
        elif position_type == "fourier":
            self.position_bias = FourierPositionEmbedding(coord_size=latent_size, hidden_size=in_ch)
        elif position_type == "fourier+learned":
            self.position_bias_fourier = FourierPositionEmbedding(coord_size=latent_size, hidden_size=in_ch)
            self.position_bias_learned = PositionEmbedding2DLearned(n_row=latent_size, feats_dim=in_ch)
        else:
            raise NotImplementedError()

    def _position_bias(self, h, w, device):
        four = learned = None
        ctx = {}
        if self.position_type in ("fourier", "fourier+learned"):
            mod = self.position_bias if self.position_type == "fourier" else self.position_bias_fourier
            four, pre, coord = mod.bias_hwc(device)
            ctx.update(pre=pre, coord=coord)
        if self.position_type == "fourier+learned":
            learned = self.position_bias_learned.bias_hwc(h, w)
        if four is None and learned is None:
            return None, ctx
        bias = four if learned is None else (learned if four is None else four + learned)
        return bias.contiguous().float(), ctx

    def fwd(self, z, tape, grain_indices=None):
        """z NHWC [B,h,w,in_ch] -> rec NHWC [B,H,W,out_ch padded]"""
        b, hh, ww, _ = z.shape
        bias, pctx = self._position_bias(hh, ww, z.device)
        h = K.add_bias_bcast(z, bias) if bias is not None else z
        h = self.conv_in.fwd(h, _child(tape, "conv_in"))
        h = self.mid.block_1.fwd(h, _child(tape, "m1"))
        h = self.mid.attn_1.fwd(h, _child(tape, "ma"))
        h = self.mid.block_2.fwd(h, _child(tape, "m2"))
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block].fwd(h, _child(tape, f"u{i_level}b{i_block}"))
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].fwd(h, _child(tape, f"u{i_level}a{i_block}"))
            if i_level != 0:
                h = lvl.upsample.fwd(h, _child(tape, f"u{i_level}us"))
        if tape is not None:
            tape.s["pctx"] = pctx
        if self.give_pre_end:                         # DecoderPositional.py:139-140: the features before norm_out / conv_out
            return h
        return norm_swish_conv(self.norm_out, self.conv_out, h, tape, "no", "co")

    def bwd(self, g, tape):
        if not self.give_pre_end:
            g = self.conv_out.bwd(g, tape.child("co"))
            g = self.norm_out.bwd(g, tape.child("no"))
        for i_level in range(self.num_resolutions):
            lvl = self.up[i_level]
            if i_level != 0:
                g = lvl.upsample.bwd(g, tape.child(f"u{i_level}us"))
            for i_block in reversed(range(self.num_res_blocks + 1)):
                if len(lvl.attn) > 0:
                    g = lvl.attn[i_block].bwd(g, tape.child(f"u{i_level}a{i_block}"))
                g = lvl.block[i_block].bwd(g, tape.child(f"u{i_level}b{i_block}"))
        g = self.mid.block_2.bwd(g, tape.child("m2"))
        g = self.mid.attn_1.bwd(g, tape.child("ma"))
        g = self.mid.block_1.bwd(g, tape.child("m1"))
        g = self.conv_in.bwd(g, tape.child("conv_in"))
        # position-bias gradients: batch reduction on the GPU, tiny parameter maps on the host side
        if self.position_type in ("fourier", "fourier+learned"):
            b, hh, ww, c = g.shape
            gsum = K.sum_batch(g, torch.zeros(hh, ww, c, dtype=torch.float32, device=g.device))
            pctx = tape.s["pctx"]
            mod = self.position_bias if self.position_type == "fourier" else self.position_bias_fourier
            mod.accumulate_grad(gsum, pctx["pre"], pctx["coord"])
            if self.position_type == "fourier+learned":
                self.position_bias_learned.accumulate_grad(gsum)
        return g

    def _fwd_nchw(self, x, tape):
        y = self.fwd(to_nhwc(x, rt.compute_dtype()), tape)
        return K.nhwc_pad_to_nchw(y, y.shape[-1] if self.give_pre_end else self.out_ch)

    def _bwd_nchw(self, dy, tape, in_dtype):
        cd = rt.compute_dtype()
        n_out = dy.shape[1] if self.give_pre_end else self.out_ch
        g = K.nchw_to_nhwc_pad(dy.contiguous().float(), K.vec(cd) * -(-n_out // K.vec(cd)), cd)
        dx = self.bwd(g, tape)
        return to_nchw(K.cast(dx, in_dtype))

    def forward(self, h, grain_indices=None):
        return super().forward(h)


# ---------------------------------------------------------------------------------------------
class DualGrainVQModel(nn.Module):
    """dqvae_dual_entropy.py:65-262.  The Lightning surface (training_step / validation_step /
    configure_optimizers / log / attributes set by train.py) is kept; the trainer is
    dynamicvectorquantization_amd.trainer (pytorch_lightning is not a dependency)."""

    USES_ENTROPY = True          # dqvae_dual_entropy: per-patch entropy feeds the fixed router; the *_feat models have none
    N_GRAINS = 2
    GRAPH_SAFE = True            # the training step has a static launch sequence: the Trainer may record it as a hipGraph

    def __init__(self, encoderconfig, decoderconfig, lossconfig, vqconfig, quant_before_dim, quant_after_dim,
                 quant_sample_temperature=0., ckpt_path=None, ignore_keys=[], image_key="image", monitor=None,
                 warmup_epochs=0, loss_with_epoch=True, scheduler_type="linear-warmup_cosine-decay",
                 entropy_patch_size=16, image_size=256):
        super().__init__()
        self.image_key = image_key
        self.encoder = instantiate_from_config(encoderconfig)
        self.decoder = instantiate_from_config(decoderconfig)
        self.loss = instantiate_from_config(lossconfig)
        self.quantize = instantiate_from_config(vqconfig)
        self.quant_conv = Conv2d(quant_before_dim, quant_after_dim, 1)
        self.post_quant_conv = Conv2d(quant_after_dim, quant_before_dim, 1)
        self.quant_sample_temperature = quant_sample_temperature
        self.entropy_patch_size, self.image_size = entropy_patch_size, image_size
        self.feature_routed = bool(getattr(self.encoder, "feature_routed", False))
        if self.USES_ENTROPY:
            self.entropy_calculation = Entropy(entropy_patch_size, image_size, image_size)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        if monitor is not None:
            self.monitor = monitor
        self.warmup_epochs, self.loss_with_epoch, self.scheduler_type = warmup_epochs, loss_with_epoch, scheduler_type
        # trainer-provided state (train.py:243-267)
        self.learning_rate, self.min_learning_rate = 0.0, 0.0
        self.training_steps, self.steps_per_epoch, self.max_epoch = 1, 1, 1
        self.current_epoch, self.global_step = 0, 0
        self._logged = {}
        self._grad_hook = None        # set by the Trainer under data parallelism (gradient exchange overlapped with backward)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            for ik in ignore_keys:
                if k.startswith(ik):
                    print("Deleting key {} from state_dict.".format(k))
                    del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        rt.bump_weights_epoch()
        return out

    # -- fused AE core (NHWC) ---------------------------------------------------------------------
    def _threshold(self):
        return self.encoder.router.fine_grain_threshold

    def ae_fwd(self, x, tape):
        """x NCHW fp32 -> dict(rec NCHW fp32, qloss, codes, grain, gate, entropy)"""
        if self.feature_routed:
            ent = None
            h_dual, mask, routing = self.encoder.fwd(x, None, _child(tape, "enc"))
            grain, gate_out = routing.indices, routing.gate.detach()          # gate: fp32 [B,S,hc,wc], differentiable
        else:
            ent, gate = K.patch_entropy_gate(x, self.entropy_patch_size, self._threshold())
            grain = gate[..., 1].contiguous()
            gate_out = gate.permute(0, 3, 1, 2)
            h_dual, mask, _ = self.encoder.fwd(x, grain, _child(tape, "enc"))
        h = self.quant_conv.fwd(h_dual, _child(tape, "qc"))
        xq, qloss, codes = self.quantize.fwd(h, mask, _child(tape, "vq"))
        z = self.post_quant_conv.fwd(xq, _child(tape, "pqc"))
        rec_p = self.decoder.fwd(z, _child(tape, "dec"))
        rec = K.nhwc_pad_to_nchw(rec_p, self.decoder.out_ch)
        if tape is not None:
            self._publish_last_layer_wgrad(tape.child("dec").child("co"))
        return {"rec": rec, "qloss": qloss, "codes": codes, "grain": grain, "gate": gate_out,
                "entropy": ent, "quant": xq, "mask": mask}

    def _publish_last_layer_wgrad(self, t_co):
        """calculate_adaptive_weight (vqperceptual_multidisc.py:97-107) differentiates two scalars w.r.t. the decoder's
        last conv weight.  Both are functions of the reconstruction only, so each gradient is ONE wgrad of conv_out
        with the corresponding d/d(rec) -- this closure is what the loss module calls (no autograd graph exists)."""
        conv = self.decoder.conv_out

        def wgrad(g_rec_p):
            buf = torch.zeros(conv.weight.shape, dtype=torch.float32, device=g_rec_p.device)
            K.conv2d_wgrad_oihw(t_co.s["d"], t_co.s["x"], g_rec_p, conv.in_channels, conv.out_channels, buf, None,
                                gn_ss=t_co.s.get("gn_ss"))
            return buf

        def wgrad_pair(g_a, g_b):
            """both gradients from ONE pass over the decoder's last activation (1 GB at bs 64, 256^2: the weight-gradient kernel of
            this 3-channel conv is bound by reading it): the two 3-channel image gradients ride side by side in the 8-channel
            padded tensor (channels 0-2 / 3-5), the result rows are split afterwards"""
            co = conv.out_channels
            if g_a.dtype != torch.bfloat16 or g_a.shape[-1] != 8 or 2 * co > 8:
                return wgrad(g_a), wgrad(g_b)
            both = K.channel_shift_add8(g_a, g_b, co)
            buf = torch.zeros((2 * co,) + tuple(conv.weight.shape[1:]), dtype=torch.float32, device=g_a.device)
            K.conv2d_wgrad_oihw(t_co.s["d"], t_co.s["x"], both, conv.in_channels, 2 * co, buf, None, gn_ss=t_co.s.get("gn_ss"))
            return buf[:co], buf[co:]

        conv.weight._dvq_wgrad = wgrad
        conv.weight._dvq_wgrad_pair = wgrad_pair

    def ae_bwd(self, g_rec, g_qloss, tape, g_gate=None):
        cd = rt.compute_dtype()
        g = K.nchw_to_nhwc_pad(g_rec, K.vec(cd) * -(-self.decoder.out_ch // K.vec(cd)), cd)
        g = self.decoder.bwd(g, tape.child("dec"))
        g = self.post_quant_conv.bwd(g, tape.child("pqc"))
        g = self.quantize.bwd(g, g_qloss, tape.child("vq"))
        g = self.quant_conv.bwd(g, tape.child("qc"))
        if self._grad_hook is not None:            # data parallel: decoder-side gradients are final -> start their all-reduce
            self._grad_hook(list(self.decoder.parameters()) + list(self.quant_conv.parameters()) +
                            list(self.post_quant_conv.parameters()))
        self.encoder.bwd(g, tape.child("enc"), g_gate)

    # -- reference API ------------------------------------------------------------------------------
    def encode(self, x):
        x_entropy = None
        if self.USES_ENTROPY:
            with torch.no_grad():
                x_entropy = self.entropy_calculation(x)
        h_dict = self.encoder(x, x_entropy)
        h = self.quant_conv(h_dict[self.encoder.OUT_KEY])
        quant, emb_loss, info = self.quantize(x=h, temp=self.quant_sample_temperature, codebook_mask=h_dict["codebook_mask"])
        if self.USES_ENTROPY:
            return quant, emb_loss, info, h_dict["indices"], h_dict["gate"], x_entropy
        return quant, emb_loss, info, h_dict["indices"], h_dict["gate"]

    def decode(self, quant, grain_indices=None):
        return self.decoder(self.post_quant_conv(quant), grain_indices)

    def _forward5(self, input):
        x = input.contiguous().float() if input.dtype != torch.float32 or not input.is_contiguous() else input
        params = [p for p in self.ae_parameters() if p.requires_grad]
        return _AEFn.apply(self, torch.is_grad_enabled() and len(params) > 0, x, *params)

    def forward(self, input):
        """-> (dec, diff, grain_indices, gate, x_entropy) [entropy-routed] / (dec, diff, grain_indices, gate) [feature-routed];
        one fused autograd node."""
        out = self._forward5(input)
        return out if self.USES_ENTROPY else out[:4]

    def ae_parameters(self):
        return (list(self.encoder.parameters()) + list(self.decoder.parameters()) + list(self.quantize.parameters()) +
                list(self.quant_conv.parameters()) + list(self.post_quant_conv.parameters()))

    def get_input(self, batch, k):
        x = batch[k]
        if len(x.shape) == 3:
            x = x[..., None]
        if x.size(1) != 3:
            x = x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()
        return x

    # logging shim (Lightning's self.log / self.log_dict)
    def log(self, name, value, **kw):
        self._logged[name] = value

    def log_dict(self, d, **kw):
        self._logged.update(d)

    reuse_generator_forward = False   # True: the discriminator step reuses the generator step's reconstruction
                                      # (one autoencoder forward + one EMA update per batch; NOT the reference's schedule)

    def training_step(self, batch, batch_idx, optimizer_idx):
        x = self.get_input(batch, self.image_key)
        prefetch = getattr(self.loss, "prefetch_targets", None)
        if prefetch is not None:      # target-only work of the loss starts on the side stream, beside the forward below
            prefetch(x, optimizer_idx, self.current_epoch if self.loss_with_epoch else self.global_step)
        if optimizer_idx == 1 and self.reuse_generator_forward and getattr(self, "_gen_out", None) is not None:
            xrec, qloss, indices, gate, x_entropy = self._gen_out
        elif optimizer_idx == 1:
            with torch.no_grad():     # the discriminator loss detaches the reconstruction: same values, no tape kept
                xrec, qloss, indices, gate, x_entropy = self._forward5(x)
        else:
            xrec, qloss, indices, gate, x_entropy = self._forward5(x)
            if self.reuse_generator_forward:
                self._gen_out = (xrec.detach(), qloss.detach(), indices, gate, x_entropy)
        step = self.current_epoch if self.loss_with_epoch else self.global_step
        if optimizer_idx == 0:
            aeloss, log_dict_ae = self.loss(qloss, x, xrec, optimizer_idx, step, last_layer=self.get_last_layer(),
                                            split="train", gate=gate)
            self.log("train_aeloss", aeloss)
            self._log_ratios("train", indices)
            self.log("train_rec_loss", log_dict_ae.pop("train_rec_loss"))
            self.log_dict(log_dict_ae)
            return aeloss
        if optimizer_idx == 1:
            discloss, log_dict_disc = self.loss(qloss, x, xrec, optimizer_idx, step, last_layer=self.get_last_layer(),
                                                split="train")
            self.log("train_discloss", discloss)
            self.log_dict(log_dict_disc)
            return discloss

    def _log_ratios(self, split, indices):
        n = indices.size(0) * indices.size(1) * indices.size(2)
        if self.N_GRAINS == 2:
            self.log(f"{split}_fine_ratio", indices.sum() / n)
        else:       # dqvae_triple_feat.py:113-115 (the reference's spelling)
            self.log(f"{split}_fine_radio", (indices == 2).sum() / n)
            self.log(f"{split}_median_radio", (indices == 1).sum() / n)

    def validation_step(self, batch, batch_idx):
        x = self.get_input(batch, self.image_key)
        with torch.no_grad():
            xrec, qloss, indices, gate, x_entropy = self._forward5(x)
            self._log_ratios("val", indices)
            step = self.current_epoch if self.loss_with_epoch else self.global_step
            aeloss, log_dict_ae = self.loss(qloss, x, xrec, 0, step, last_layer=self.get_last_layer(), split="val", gate=gate)
            discloss, log_dict_disc = self.loss(qloss, x, xrec, 1, step, last_layer=self.get_last_layer(), split="val")
        self.log("val_rec_loss", log_dict_ae.pop("val_rec_loss"))
        self.log("val_aeloss", aeloss)
        self.log_dict(log_dict_ae)
        self.log_dict(log_dict_disc)
        return self.log_dict

    def configure_optimizers(self):
        from .trainer import HipAdam, scheduler_linear_warmup, scheduler_linear_warmup_cosine_decay
        lr = self.learning_rate
        opt_ae = HipAdam(self.ae_parameters(), lr=lr, betas=(0.5, 0.9))
        disc = getattr(self.loss, "discriminator", None)
        if getattr(self.loss, "disc_factor", 0) == 0:
            disc = None     # AE-only objective: no discriminator pass, no second forward
        opt_disc = HipAdam(list(disc.parameters()), lr=lr, betas=(0.5, 0.9)) if disc is not None else None
        warmup_steps = self.steps_per_epoch * self.warmup_epochs
        if self.scheduler_type == "linear-warmup":
            fn = scheduler_linear_warmup(warmup_steps)
        elif self.scheduler_type == "linear-warmup_cosine-decay":
            multipler_min = self.min_learning_rate / self.learning_rate if self.learning_rate else 0.0
            fn = scheduler_linear_warmup_cosine_decay(warmup_steps, self.training_steps, multipler_min)
        else:
            raise NotImplementedError()
        opts = [opt_ae] + ([opt_disc] if opt_disc is not None else [])
        scheds = [{"scheduler": torch.optim.lr_scheduler.LambdaLR(o, fn), "interval": "step", "frequency": 1} for o in opts]
        return opts, scheds

    def get_last_layer(self):
        return self.decoder.conv_out.weight

    def graph_signature(self):
        """host-side state that decides which kernels a training step launches (Trainer step capture); None = do not capture"""
        cb = getattr(self.quantize, "codebook", None)
        if getattr(cb, "restart_perm", None) is not None or getattr(self.encoder, "gumbel_exponential", None) is not None:
            return None                      # injected test noise lives on the host
        if any(isinstance(m, ResnetBlock) and m.dropout.p > 0.0 for m in self.modules()):
            return None                      # dropout seeds are drawn on the host per call (a replay would repeat the masks)
        if getattr(getattr(self.loss, "perceptual_loss", None), "lin_dropout", False):
            return None                      # opt-in LPIPS lin-layer dropout: its seeds are drawn on the host per call
        from .layers import ActNorm
        if any(isinstance(m, ActNorm) and self.training and not m.inited() for m in self.modules()):
            return None                      # ActNorm's data-dependent initialisation reads the batch on the host: first step(s) eager
        step = self.current_epoch if self.loss_with_epoch else self.global_step
        disc_on = None
        if hasattr(self.loss, "discriminator_iter_start"):
            disc_on = bool(step >= self.loss.discriminator_iter_start)
        return (disc_on, bool(self.reuse_generator_forward), self.feature_routed, self.N_GRAINS)

    def get_code_emb_with_depth(self, code):
        return self.quantize.get_codebook_entry(code)


class DualGrainFeatVQModel(DualGrainVQModel):
    """models/stage1_dynamic/dqvae_dual_feat.py: the same autoencoder without the entropy branch (feature router);
    forward -> (dec, diff, grain_indices, gate)"""
    USES_ENTROPY = False


class TripleGrainVQModel(DualGrainVQModel):
    """models/stage1_dynamic/dqvae_triple_feat.py:22-199: three grains (F = 32/16/8), TripleGrainFeatureRouter,
    forward -> (dec, diff, grain_indices, gate)"""
    USES_ENTROPY = False
    N_GRAINS = 3


class _AEFn(torch.autograd.Function):
    """The whole autoencoder as one autograd node: forward fills a Tape, backward walks it."""

    @staticmethod
    def forward(ctx, model, want_grad, x, *params):
        ctx.model, ctx.n = model, len(params)
        ctx.tape = Tape() if want_grad else None    # grad mode is off inside Function.forward
        with torch.no_grad():
            out = model.ae_fwd(x, ctx.tape)
        model._last = out
        nd = ["grain"] + (["gate", "entropy"] if not model.feature_routed else [])
        for k in nd:
            ctx.mark_non_differentiable(out[k])
        ctx.set_materialize_grads(False)
        return out["rec"], out["qloss"], out["grain"], out["gate"], out["entropy"]

    @staticmethod
    def backward(ctx, g_rec, g_qloss, _g_grain=None, g_gate=None, _g_ent=None):
        with torch.no_grad():
            if g_rec is None:
                g_rec = torch.zeros_like(ctx.model._last["rec"])
            if g_qloss is None:
                g_qloss = torch.zeros((), device=g_rec.device)
            ctx.model.ae_bwd(g_rec.contiguous().float(), g_qloss, ctx.tape, g_gate if ctx.model.feature_routed else None)
        return (None, None, None) + (None,) * ctx.n
