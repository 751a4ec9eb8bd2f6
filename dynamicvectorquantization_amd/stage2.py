"""Stage-2 (DQ-Transformer) input plumbing on the HIP path: code maps <-> sequences.

Mirrors /root/reference/modules/dynamic_modules/permuter.py:6-135 (DualGrainSeperatePermuter) and
modules/dynamic_modules/label_provider.py:4-92 (SOS providers).  The permutation is integer compaction / scatter work done
by dvq_permute_dual / dvq_permute_dual_back (one workgroup per image); results are bit-exact with the reference, including
its sequential edge semantics (missing EOS, duplicate positions).  The transformer itself (`stackgpt.StackGPT`: fused causal
attention, LayerNorm / GELU / embedding / cross-entropy kernels, K/V-cached graph-replayed sampling) and `Dualformer` (below) run on
the same library.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import check
from .config import instantiate_from_config
from .kernels import lib


import threading as _threading

_SAMPLER_LANE = _threading.local()          # .index: 0 = the sequential sampler; 1.. = a lane of Dualformer.sample_many

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


class _SamplerMixinBase:
    """filled in below (_SamplerMixin): the sampler's methods are attached to Dualformer after both are defined"""


def disabled_train(self, mode=True):
    """models/stage2/utils.py:20-23: the frozen first stage never leaves eval mode"""
    return self


class Dualformer(_SamplerMixinBase, nn.Module):
    """models/stage2_dynamic/dqtransformer_uncond_entropy.py:15-234: the stage-2 LightningModule surface without Lightning.
    A frozen DQ-VAE (HIP path) encodes images to codes + grain map, the permuter kernel turns them into the coarse / fine
    streams, StackGPT (HIP path) is trained with teacher forcing; AdamW(betas .9/.95) with the reference's decay /
    no-decay parameter split."""

    def __init__(self, transformer_config, first_stage_config, uncond_stage_config=None, permuter_config=None,
                 content_loss_weight=1.0, position_loss_weight=1.0, activate_sos_for_fine_sequence=True, weight_decay=0.01,
                 warmup_epochs=0, monitor=None, ckpt_path=None, ignore_keys=[]):
        super().__init__()
        self.first_stage_key, self.cond_stage_key = "image", "image"
        self.content_loss_weight, self.position_loss_weight = content_loss_weight, position_loss_weight
        self.init_first_stage_from_ckpt(first_stage_config)
        self.permuter = instantiate_from_config(config=permuter_config)
        self.transformer = instantiate_from_config(config=transformer_config)
        self.cond_stage_model = instantiate_from_config(config=uncond_stage_config)
        self.weight_decay, self.warmup_epochs = weight_decay, warmup_epochs
        if monitor is not None:
            self.monitor = monitor
        self.activate_sos_for_fine_sequence = activate_sos_for_fine_sequence
        self.activate_segment = transformer_config["params"]["segment_size"] > 0
        pp, up = permuter_config["params"], uncond_stage_config["params"]
        self.content_pad_code, self.content_eos_code = pp["content_pad_code"], pp["content_eos_code"]
        self.content_sos_code = up.get("coarse_sos")
        self.coarse_position_eos_code, self.coarse_position_pad_code = pp["coarse_position_eos_code"], pp["coarse_position_pad_code"]
        self.fine_position_sos_code = up.get("fine_pos_sos")
        self.fine_position_eos_code, self.fine_position_pad_code = pp["fine_position_eos_code"], pp["fine_position_pad_code"]
        self.hw1, self.fine_hw = pp["coarse_hw"], pp["fine_hw"]
        self.hw2 = self.fine_hw // self.hw1
        self.fine_position_order = pp["fine_position_order"]
        self.max_coarse_postion_idx = int(self.hw1 ** 2) - 1
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        # trainer-provided state (train.py:243-267)
        self.learning_rate, self.min_learning_rate = 0.0, 0.0
        self.training_steps, self.steps_per_epoch, self.max_epoch = 1, 1, 1
        self.current_epoch, self.global_step = 0, 0
        self._logged = {}

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            for ik in ignore_keys:
                if k.startswith(ik):
                    del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def init_first_stage_from_ckpt(self, config):
        model = instantiate_from_config(config).eval()
        for p in model.parameters():
            p.requires_grad = False
        model.train = disabled_train.__get__(model)
        self.first_stage_model = model

    def log(self, name, value, **kw):
        self._logged[name] = value

    def configure_optimizers(self):
        """AdamW groups of dqtransformer_uncond_entropy.py:92-143: Linear weights decay; biases, LayerNorm / Embedding weights
        and pos_emb do not"""
        from .layers import Linear
        from .stackgpt import LayerNorm
        from .trainer import HipAdam, scheduler_linear_warmup_cosine_decay
        decay, no_decay = set(), set()
        for mn, m in self.transformer.named_modules():
            for pn, _ in m.named_parameters(recurse=False):
                fpn = f"{mn}.{pn}" if mn else pn
                if pn.endswith("bias"):
                    no_decay.add(fpn)
                elif pn.endswith("weight") and isinstance(m, Linear):
                    decay.add(fpn)
                elif pn.endswith("weight") and isinstance(m, (LayerNorm, nn.Embedding)):
                    no_decay.add(fpn)
        no_decay.add("pos_emb")
        param_dict = {pn: p for pn, p in self.transformer.named_parameters()}
        assert len(decay & no_decay) == 0 and len(param_dict.keys() - (decay | no_decay)) == 0
        groups = [{"params": [param_dict[pn] for pn in sorted(decay)], "weight_decay": self.weight_decay},
                  {"params": [param_dict[pn] for pn in sorted(no_decay)], "weight_decay": 0.0}]
        opt = HipAdam(groups, lr=self.learning_rate, betas=(0.9, 0.95))
        warmup_steps = self.steps_per_epoch * self.warmup_epochs
        mmin = self.min_learning_rate / self.learning_rate if self.learning_rate else 0.0
        fn = scheduler_linear_warmup_cosine_decay(warmup_steps, self.training_steps, mmin)
        return [opt], [{"scheduler": torch.optim.lr_scheduler.LambdaLR(opt, fn), "interval": "step", "frequency": 1}]

    def get_input(self, batch, k):
        x = batch[k]
        if len(x.shape) == 3:
            x = x[..., None]
        if x.size(1) != 3 and len(x.shape) == 4:
            x = x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()
        return x

    def get_xc(self, batch, N=None):
        x, c = self.get_input(batch, self.first_stage_key), self.get_input(batch, self.cond_stage_key)
        if N is not None:
            x, c = x[:N], c[:N]
        return x, c

    @torch.no_grad()
    def encode_to_c(self, c):
        return self.cond_stage_model.encode(c)

    @torch.no_grad()
    def encode_to_z(self, x):
        enc = self.first_stage_model.encode(x)
        quant, info, grain_indices = enc[0], enc[2], enc[3]
        return quant, self.permuter(indices=info[2], grain_indices=grain_indices)

    @torch.no_grad()
    def decode_to_img(self, coarse_content, fine_content, coarse_position, fine_position):
        idx = self.permuter.forward_back(coarse_content, fine_content, coarse_position, fine_position)
        quant = self.first_stage_model.get_code_emb_with_depth(idx)
        return self.first_stage_model.decode(quant.permute(0, 3, 1, 2))

    def teacher_forcing_inputs(self, z_out, c):
        """SOS-prefixed streams and their shifted targets (dqtransformer_uncond_entropy.py:183-207)"""
        c_coarse, c_fine, c_pos_coarse, c_pos_fine, c_seg_coarse, c_seg_fine = c
        cc = torch.cat([c_coarse, z_out["coarse_content"]], dim=1)
        cp = torch.cat([c_pos_coarse, z_out["coarse_position"]], dim=1)
        cs = torch.cat([c_seg_coarse, z_out["coarse_segment"]], dim=1) if c_seg_coarse is not None else z_out["coarse_segment"]
        if self.activate_sos_for_fine_sequence:
            fc = torch.cat([c_fine, z_out["fine_content"]], dim=1)
            fp = torch.cat([c_pos_fine, z_out["fine_position"]], dim=1)
            fs = torch.cat([c_seg_fine, z_out["fine_segment"]], dim=1) if c_seg_fine is not None else z_out["fine_segment"]
        else:
            fc, fp, fs = z_out["fine_content"], z_out["fine_position"], z_out["fine_segment"]
        return dict(coarse_content=cc, fine_content=fc, coarse_position=cp, fine_position=fp, coarse_seg=cs, fine_seg=fs,
                    content_target=torch.cat([cc, fc], dim=1)[:, 1:], coarse_position_target=cp[:, 1:], fine_position_target=fp)

    def forward(self, x, c):
        _, z_out = self.encode_to_z(x)
        return self.transformer(**self.teacher_forcing_inputs(z_out, self.encode_to_c(c)))

    def shared_step(self, batch, batch_idx):
        x, c = self.get_xc(batch)
        return self(x, c)

    def _step(self, batch, batch_idx, split):
        out = self.shared_step(batch, batch_idx)
        total = self.content_loss_weight * out["content_loss"] + self.position_loss_weight * out["position_loss"]
        self.log(f"{split}_content_loss", out["content_loss"].detach())
        self.log(f"{split}_position_loss", out["position_loss"].detach())
        self.log(f"{split}_coarse_position_loss", out["coarse_position_loss"].detach())
        self.log(f"{split}_fine_position_loss", out["fine_position_loss"].detach())
        self.log(f"{split}_loss", total.detach())
        return total

    def training_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            return self._step(batch, batch_idx, "val")


# ---- sampling (dqtransformer_uncond_entropy.py:302-561, models/stage2/utils.py:22-40) ---------------------------------------
def top_k_logits(logits, k):
    v, _ = torch.topk(logits, k)
    # same values as the reference's `out[out < v[..., [-1]]] = -inf` without the host sync of boolean-mask assignment
    return torch.where(logits < v[..., -1:], torch.full_like(logits, -float("Inf")), logits)


def top_p_logits(probs, p):
    sorted_probs, sorted_indices = torch.sort(probs, dim=-1, descending=True)
    remove = torch.cumsum(sorted_probs, dim=-1) >= p
    remove[..., 1:] = remove[..., :-1].clone()
    remove[..., 0] = 0
    probs = probs.masked_fill(remove.scatter(-1, sorted_indices, remove), 0.0)
    return probs / torch.sum(probs, dim=-1, keepdim=True)


def _mask_rows(logits, finished, forbid, keep_code, pad_code):
    """unfinished rows: -inf where `forbid`, except `keep_code` which keeps its logit; finished rows: only `pad_code` survives"""
    neg = torch.full_like(logits, -float("Inf"))
    out = torch.where(forbid, neg, logits)
    if keep_code is not None:
        out[:, keep_code] = logits[:, keep_code]
    only_pad = neg.clone()
    only_pad[:, pad_code] = logits[:, pad_code]
    return torch.where(finished.bool().view(-1, 1), only_pad, out)


class _SamplerMixin:
    """the constrained ancestral sampler of Dualformer, batched over rows (no per-sample Python loops)"""

    def avoid_repeat_or_enforce_pad_for_coarse_position(self, logits, sampled_position, flag):
        forbid = torch.zeros_like(logits, dtype=torch.bool)
        forbid.scatter_(1, sampled_position, True)                      # <sos> and the coarse positions already drawn
        forbid[:, self.coarse_position_pad_code] = True
        forbid[:, self.max_coarse_postion_idx:] = True                 # the reference forbids index hw1^2 - 1 as well
        return _mask_rows(logits, flag, forbid, self.coarse_position_eos_code, self.coarse_position_pad_code)

    def avoid_repeat_or_enforce_pad_for_fine_position(self, logits, sampled_position, flag):
        forbid = torch.zeros_like(logits, dtype=torch.bool)
        forbid.scatter_(1, sampled_position, True)
        forbid[:, self.fine_position_pad_code] = True
        out = _mask_rows(logits, flag, forbid, self.fine_position_eos_code, self.fine_position_pad_code)
        live = ~flag.bool().view(-1)
        col = out[:, self.fine_position_sos_code]                      # applied after <eos> was restored (reference order)
        out[:, self.fine_position_sos_code] = torch.where(live, torch.full_like(col, -float("Inf")), col)
        return out

    def avoid_special_or_enforce_pad_for_content(self, logits, flag):
        forbid = torch.zeros_like(logits, dtype=torch.bool)
        for code in (self.content_pad_code, self.content_eos_code, self.content_sos_code):
            forbid[:, code] = True
        return _mask_rows(logits, flag, forbid, None, self.content_pad_code)

    def _coarse_cells_drawn(self, coarse_position):
        """[B, hw1, hw1] int64: 1 where a coarse position was sampled before the row's <eos> (column 0 is <sos>)"""
        pos = coarse_position[:, 1:]
        alive = torch.cumsum((pos == self.coarse_position_eos_code).long(), dim=1) == 0
        n = self.hw1 * self.hw1
        drawn = torch.zeros(pos.shape[0], n + 1, dtype=torch.long, device=pos.device)
        drawn.scatter_(1, torch.where(alive, pos.clamp(0, n - 1), torch.full_like(pos, n)), 1)
        return drawn[:, :n].reshape(-1, self.hw1, self.hw1)

    def _fine_prefix(self, coarse_position):
        """first column of the transferred fine-position stream: the fine <sos> position code (unconditional model)"""
        return torch.full_like(coarse_position[:, :1], self.fine_position_sos_code)

    def _fine_positions_of(self, cell_flags, coarse_position):
        """fine-position stream (<eos>-terminated, padded) of the cells flagged 1, in the permuter's fine order"""
        dummy = torch.zeros(cell_flags.shape[0], self.fine_hw, self.fine_hw, dtype=torch.long, device=cell_flags.device)
        fp = self.permuter(indices=dummy, grain_indices=cell_flags)["fine_position"]
        if self.activate_sos_for_fine_sequence:
            fp = torch.cat([self._fine_prefix(coarse_position), fp], dim=1)
        return fp

    def transfer_sampled_coarse_position_to_sampled_fine_position(self, coarse_position):
        return self._fine_positions_of(self._coarse_cells_drawn(coarse_position), coarse_position)

    def transfer_sampled_coarse_position_to_remain_fine_position(self, coarse_position):
        return self._fine_positions_of(1 - self._coarse_cells_drawn(coarse_position), coarse_position)

    def _draw_rule(self, logits2d, temperature, sample, k, p, rule):
        """one token per row under constraint `rule` = (kind, sampled positions or None, done flags): ONE fused launch
        (kernels.sample_constrained: mask rules + top-k / top-p + softmax + multinomial / top-1) when the logits are on the device;
        DVQ_SAMPLER=torch (or a vocabulary beyond the kernel's 2048 columns) keeps the op-by-op path the golden tests pin"""
        kind, sampled, done = rule
        if logits2d.is_cuda and logits2d.shape[1] <= 2048 and logits2d.dtype in (torch.float32, torch.bfloat16) and \
                os.environ.get("DVQ_SAMPLER", "fused") != "torch":
            # {seed, counter} of the kernel's counter-based generator, keyed by torch's seed: a later torch.manual_seed() /
            # seed_everything() restarts the stream (same seed -> same samples, like the op-by-op path's torch generator)
            seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
            lane = int(getattr(_SAMPLER_LANE, "index", 0))      # concurrent sampling (sample_many): one generator stream per lane
            if lane:
                seed = (seed * 0x9E3779B1 + lane) & 0x7FFFFFFFFFFFFFFF
                states = self.__dict__.setdefault("_sampler_lane_states", {})
                st = states.get(lane)
                if st is None or st[1].device != logits2d.device or st[0] != seed:
                    st = states[lane] = (seed, torch.tensor([seed, 0], dtype=torch.int64, device=logits2d.device))
                st = st[1]
            else:
                st = self.__dict__.get("_sampler_state")
                if st is None or st.device != logits2d.device or self.__dict__.get("_sampler_seed") != seed:
                    st = torch.tensor([seed, 0], dtype=torch.int64, device=logits2d.device)
                    self.__dict__["_sampler_state"] = st
                    self.__dict__["_sampler_seed"] = seed
            kw = self._fused_rule(kind)
            if kind != "content":
                kw["forbid_idx"] = sampled
            return K.sample_constrained(logits2d, temperature, state=st, finished=done, top_k=k, top_p=p, sample=sample, **kw)
        fn = {"coarse_pos": lambda lg: self.avoid_repeat_or_enforce_pad_for_coarse_position(lg, sampled, done),
              "fine_pos": lambda lg: self.avoid_repeat_or_enforce_pad_for_fine_position(lg, sampled, done),
              "content": lambda lg: self.avoid_special_or_enforce_pad_for_content(lg, done)}[kind]
        return self._draw(logits2d.unsqueeze(1), temperature, sample, k, p, fn)

    def _fused_rule(self, kind):
        """the three mask rules (avoid_repeat_or_enforce_pad_for_* / avoid_special_or_enforce_pad_for_content above) as arguments of
        kernels.sample_constrained"""
        if kind == "coarse_pos":
            return dict(pad_code=self.coarse_position_pad_code, forbid_from=self.max_coarse_postion_idx,
                        forbid_codes=(self.coarse_position_pad_code,), keep_code=self.coarse_position_eos_code)
        if kind == "fine_pos":
            return dict(pad_code=self.fine_position_pad_code, forbid_codes=(self.fine_position_pad_code,),
                        keep_code=self.fine_position_eos_code, late_forbid_code=self.fine_position_sos_code)
        return dict(pad_code=self.content_pad_code, forbid_codes=(self.content_pad_code, self.content_eos_code, self.content_sos_code))

    @staticmethod
    def _draw(logits, temperature, sample, k, p, constrain):
        logits = constrain(logits[:, -1, :] / temperature)
        if k is not None:
            logits = top_k_logits(logits, k)
        probs = torch.softmax(logits, dim=-1)
        if p is not None:
            probs = top_p_logits(probs, p)
        return torch.multinomial(probs, num_samples=1) if sample else torch.topk(probs, k=1, dim=-1)[1]

    @torch.no_grad()
    def _sample_cached(self, c_coarse, c_fine, c_pos_coarse, c_pos_fine, c_seg_coarse, c_seg_fine, temperature, sample, top_k, top_p,
                       top_k_pos, top_p_pos, fix_fine_position):
        """the same sampler with K/V caches: every step feeds ONE new row to each transformer instead of recomputing the whole
        prefix (the reference's O(T^2) schedule).  The content transformer's coarse rows are re-filled once when the fine
        stream starts, because the reference pairs them with different update positions in the two phases
        (stackgpt.py:263 vs :331: shifted coarse positions while sampling coarse, unshifted ones afterwards)."""
        from .stackgpt import DecodeState
        tr = self.transformer
        cpe, fpe = tr.content_coarse_pos_emb.weight, tr.content_fine_pos_emb.weight
        seg = self.activate_segment
        x_c, x_pc, x_sc = c_coarse, c_pos_coarse, c_seg_coarse
        if self.activate_sos_for_fine_sequence:
            x_f, x_pf, x_sf = c_fine, c_pos_fine, c_seg_fine
        else:
            x_f, x_pf, x_sf = c_fine[:, :0], c_pos_fine[:, :0], (c_seg_fine[:, :0] if c_seg_fine is not None else None)
        b, dev = x_c.size(0), x_c.device
        # buffers and captured per-token graphs are kept across sampling runs of the same batch size
        rows = self.hw1 * self.hw1 + self.fine_hw * self.fine_hw + 8
        pool = self.__dict__.setdefault("_decode_states", {})
        from . import runtime as _rt
        lane = int(getattr(_SAMPLER_LANE, "index", 0))        # sample_many: one state (caches + captured token-step graphs) per lane
        key = (b, rows, str(dev), str(_rt.compute_dtype()), lane)
        st = pool.get(key)
        if st is None or st.gpt is not tr:
            for k_ in [k_ for k_ in pool if k_[4] == lane]:   # one resident state per lane: the K/V caches are the big allocation
                pool.pop(k_)
            st = pool[key] = DecodeState(tr, b, rows)
        st.reset()
        zeros1 = torch.zeros(b, 1, dtype=torch.long, device=dev)
        # ---- coarse stream
        done = torch.zeros(b, 1, device=dev)
        while not torch.all(done.bool()):
            pl = st.position_rows(x_c[:, -1:], x_pc[:, -1:], cpe, None, x_sc[:, -1:] if seg else None)
            ix_pos = self._draw_rule(pl, temperature, sample, top_k_pos, top_p_pos, ("coarse_pos", x_pc, done))
            x_pc = torch.cat((x_pc, ix_pos), dim=1)
            done = done + (ix_pos == self.coarse_position_eos_code)
            cl = st.content_rows(ix_pos, cpe)
            ix = self._draw_rule(cl, temperature, sample, top_k, top_p, ("content", None, done))
            if seg:
                x_sc = torch.cat([x_sc, zeros1], dim=1)
            x_c = torch.cat((x_c, ix), dim=1)
        # ---- fine stream
        n_coarse = x_c.shape[1]
        st.reset_content()
        done = torch.zeros(b, 1, device=dev)
        fed_fine = 0
        if fix_fine_position:
            plan = self.transfer_sampled_coarse_position_to_remain_fine_position(x_pc)
            taken = None
        else:
            plan = None
            taken = self.transfer_sampled_coarse_position_to_sampled_fine_position(x_pc)
        j = 1 if self.activate_sos_for_fine_sequence else 0
        while True:
            if fix_fine_position:
                if j >= plan.size(1):
                    break
            elif torch.all(done.bool()):
                break
            # position rows that have not been fed yet: the last coarse row, then the fine rows
            pl = None
            if st.rows_pos == n_coarse - 1:
                pl = st.position_rows(x_c[:, -1:], x_pc[:, -1:], cpe, None, x_sc[:, -1:] if seg else None)
            while fed_fine < x_f.shape[1]:
                pl = st.position_rows(x_f[:, fed_fine:fed_fine + 1], x_pf[:, fed_fine:fed_fine + 1], fpe, None,
                                      x_sf[:, fed_fine:fed_fine + 1] if seg else None)
                fed_fine += 1
            if fix_fine_position:
                ix_pos = plan[:, j].unsqueeze(-1)
                j += 1
            else:
                ix_pos = self._draw_rule(pl, temperature, sample, top_k_pos, top_p_pos, ("fine_pos", taken, done))
                taken = torch.cat([taken, ix_pos], dim=1)
            x_pf = torch.cat((x_pf, ix_pos), dim=1)
            done = done + (ix_pos == self.fine_position_eos_code)
            # content rows: the coarse rows once (unshifted coarse positions), then one fine row per step
            if st.rows_con == 0:
                st.content_rows(x_pc[:, :n_coarse], cpe)
            k = st.rows_con - n_coarse                       # fine rows already in the content transformer
            cl = None
            while st.rows_con < st.rows_pos:
                cl = st.content_rows(x_pf[:, k + 1:k + 2], fpe)
                k += 1
            if cl is None:                                   # no fine <sos>: the first fine content comes from the last coarse row
                st.reset_content()
                cl = st.content_rows(x_pc[:, :n_coarse], cpe)
            ix = self._draw_rule(cl, temperature, sample, top_k, top_p, ("content", None, done))
            x_f = torch.cat((x_f, ix), dim=1)
            if seg:
                x_sf = torch.cat([x_sf, zeros1 + 1], dim=1)
        st.check()                   # the persistent token-step kernel reports a barrier time-out here: never return garbage tokens
        x_c, x_pc = x_c[:, c_coarse.shape[1]:], x_pc[:, c_pos_coarse.shape[1]:]
        if self.activate_sos_for_fine_sequence:
            x_f, x_pf = x_f[:, c_fine.shape[1]:], x_pf[:, c_fine.shape[1]:]
        return x_c, x_f, x_pc, x_pf

    @torch.no_grad()
    def sample_many(self, conds, n_streams=2, **kw):
        """sample_from_scratch for a LIST of independent batches (conds[i] = the six conditioning tensors of batch i), `n_streams`
        of them in flight at a time, each on its own HIP stream with its own K/V caches, captured token-step graphs and generator
        stream.  A token step is ~120 dependent launches of 5 - 12 us that stream a few MB each: ONE batch leaves the chip idle
        through every launch boundary and load round trip, and the reference's loop (dqtransformer_uncond_entropy.py:302-466) reads
        its `done` flags on the host every iteration.  Two lanes fill each other's gaps -- the metric is token-steps/s, the
        sampling scripts draw hundreds of batches anyway.  Each lane is a host thread (the per-iteration host read-back releases the
        GIL, so the lanes ping-pong); the first batch of every lane runs ALONE (it captures that lane's graphs: stream capture
        must not see another thread's launches).  Returns the results in the order of `conds`.  Greedy draws equal the sequential
        sampler's token for token; multinomial draws use one generator stream per lane.  The assignment is STATIC -- batch i runs
        on lane i % n_streams, each lane takes its batches in order -- so a fixed (seed, n_streams) reproduces the same samples run
        to run whatever the thread timing."""
        import threading
        n = len(conds)
        n_streams = max(1, min(int(n_streams), n))
        if n_streams == 1:
            return [self.sample_from_scratch(*c, **kw) for c in conds]
        dev = conds[0][0].device
        lanes = self.__dict__.setdefault("_sampler_lanes", {})
        streams = [lanes.setdefault((str(dev), i), torch.cuda.Stream(dev)) for i in range(n_streams)]
        results, errors = [None] * n, []
        main = torch.cuda.current_stream(dev)
        seed = torch.initial_seed()
        from . import runtime as _rt
        cd = _rt.compute_dtype()

        def run(i, lane):
            _SAMPLER_LANE.index = lane + 1
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[lane]):       # (the compute dtype is process-wide state: the lanes inherit it)
                    results[i] = self.sample_from_scratch(*conds[i], **kw)
            except BaseException as e:                  # surfaced by the caller's thread
                errors.append(e)
            finally:
                _SAMPLER_LANE.index = 0

        for s in streams:
            s.wait_stream(main)
        queues = [list(range(lane, n, n_streams)) for lane in range(n_streams)]
        b0 = int(conds[0][0].size(0))
        rows = self.hw1 * self.hw1 + self.fine_hw * self.fine_hw + 8

        def lane_is_warm(lane):
            """this lane's DecodeState exists for the geometry, was built on the current weights and holds captured token-step graphs"""
            st = self.__dict__.get("_decode_states", {}).get((b0, rows, str(dev), str(cd), lane + 1))
            return (st is not None and st.gpt is self.transformer and st._sig == st._weights_signature() and
                    (not st.use_graph or any(e.get("graph") is not None for e in st._steps.values())))

        for lane in range(n_streams):                   # graphs of a lane that has not sampled this geometry yet: alone
            if queues[lane] and not lane_is_warm(lane):
                run(queues[lane].pop(0), lane)
                streams[lane].synchronize()
        if errors:
            raise errors[0]

        def worker(lane):
            for i in queues[lane]:
                if errors:
                    return
                run(i, lane)

        threads = [threading.Thread(target=worker, args=(lane,), daemon=True) for lane in range(n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in streams:
            main.wait_stream(s)
        if errors:
            raise errors[0]
        assert torch.initial_seed() == seed
        return results

    @torch.no_grad()
    def sample_from_scratch(self, c_coarse, c_fine, c_pos_coarse, c_pos_fine, c_seg_coarse, c_seg_fine, temperature=1.0, sample=True,
                            top_k=None, top_p=None, top_k_pos=None, top_p_pos=None, process=True, fix_fine_position=False,
                            kv_cache=True):
        if kv_cache and not self.transformer.training:
            return self._sample_cached(c_coarse, c_fine, c_pos_coarse, c_pos_fine, c_seg_coarse, c_seg_fine, temperature, sample,
                                       top_k, top_p, top_k_pos, top_p_pos, fix_fine_position)
        tr = self.transformer
        x_c, x_pc, x_sc = c_coarse, c_pos_coarse, c_seg_coarse
        if self.activate_sos_for_fine_sequence:
            x_f, x_pf, x_sf = c_fine, c_pos_fine, c_seg_fine
        else:
            x_f, x_pf, x_sf = c_fine[:, :0], c_pos_fine[:, :0], c_seg_fine[:, :0]
        b, dev = x_c.size(0), x_c.device
        zeros1 = torch.zeros(b, 1, dtype=torch.long, device=dev)
        # ---- coarse stream: position, then content, until every row has drawn <eos>
        done = torch.zeros(b, 1, device=dev)
        while not torch.all(done.bool()):
            hidden, pl = tr.sample_coarse_position(coarse_content=x_c, coarse_position=x_pc, coarse_seg=x_sc)
            ix_pos = self._draw(pl, temperature, sample, top_k_pos, top_p_pos,
                                lambda lg: self.avoid_repeat_or_enforce_pad_for_coarse_position(lg, x_pc, done))
            x_pc = torch.cat((x_pc, ix_pos), dim=1)
            done = done + (ix_pos == self.coarse_position_eos_code)
            _, cl = tr.sample_coarse_content(coarse_content=None, coarse_position=x_pc, coarse_seg=None, position_hidden=hidden)
            ix = self._draw(cl, temperature, sample, top_k, top_p, lambda lg: self.avoid_special_or_enforce_pad_for_content(lg, done))
            if self.activate_segment:
                x_sc = torch.cat([x_sc, zeros1], dim=1)
            x_c = torch.cat((x_c, ix), dim=1)
        # ---- fine stream
        done = torch.zeros(b, 1, device=dev)
        if not fix_fine_position:
            taken = self.transfer_sampled_coarse_position_to_sampled_fine_position(x_pc)
            while not torch.all(done.bool()):
                hidden, pl = tr.sample_fine_position(coarse_content=x_c, fine_content=x_f, coarse_position=x_pc, fine_position=x_pf,
                                                     coarse_seg=x_sc, fine_seg=x_sf)
                ix_pos = self._draw(pl, temperature, sample, top_k_pos, top_p_pos,
                                    lambda lg: self.avoid_repeat_or_enforce_pad_for_fine_position(lg, taken, done))
                x_pf = torch.cat((x_pf, ix_pos), dim=1)
                taken = torch.cat([taken, ix_pos], dim=1)
                done = done + (ix_pos == self.fine_position_eos_code)
                _, cl = tr.sample_fine_content(coarse_content=x_c, fine_content=x_f, coarse_position=x_pc, fine_position=x_pf,
                                               coarse_seg=x_sc, fine_seg=x_sf, position_hidden=hidden)
                ix = self._draw(cl, temperature, sample, top_k, top_p, lambda lg: self.avoid_special_or_enforce_pad_for_content(lg, done))
                x_f = torch.cat((x_f, ix), dim=1)
                if self.activate_segment:
                    x_sf = torch.cat([x_sf, zeros1 + 1], dim=1)
        else:
            remain = self.transfer_sampled_coarse_position_to_remain_fine_position(x_pc)
            for j in range(remain.size(1)):
                if self.activate_sos_for_fine_sequence and j == 0:
                    continue
                ix_pos = remain[:, j].unsqueeze(-1)
                x_pf = torch.cat((x_pf, ix_pos), dim=1)
                done = done + (ix_pos == self.fine_position_eos_code)
                _, cl = tr.sample_fine_content(coarse_content=x_c, fine_content=x_f, coarse_position=x_pc, fine_position=x_pf,
                                               coarse_seg=x_sc, fine_seg=x_sf, position_hidden=None)
                ix = self._draw(cl, temperature, sample, top_k, top_p, lambda lg: self.avoid_special_or_enforce_pad_for_content(lg, done))
                x_f = torch.cat((x_f, ix), dim=1)
                if self.activate_segment:
                    x_sf = torch.cat([x_sf, zeros1 + 1], dim=1)
        x_c, x_pc = x_c[:, c_coarse.shape[1]:], x_pc[:, c_pos_coarse.shape[1]:]
        if self.activate_sos_for_fine_sequence:
            x_f, x_pf = x_f[:, c_fine.shape[1]:], x_pf[:, c_fine.shape[1]:]
        return x_c, x_f, x_pc, x_pf


for _name, _fn in list(vars(_SamplerMixin).items()):
    if not _name.startswith("__"):
        setattr(_SamplerMixinBase, _name, _fn)


class ClassDualformer(Dualformer):
    """models/stage2_dynamic/dqtransformer_class2_entropy.py: class-conditional variant.  The start tokens are the class
    label shifted above the code / position vocabularies (ClassAwareSOSProvider), so the sampler masks every id above
    <eos> instead of a single <sos> id, and the transferred fine-position stream starts with the label's position token."""

    def __init__(self, transformer_config, first_stage_config, class_cond_stage_config, permuter_config=None, **kw):
        super().__init__(transformer_config, first_stage_config, uncond_stage_config=class_cond_stage_config,
                         permuter_config=permuter_config, **kw)
        self.cond_stage_key = "class_label"
        del self.content_sos_code, self.fine_position_sos_code

    def get_xc(self, batch, N=None):
        x, c = self.get_input(batch, self.first_stage_key), batch[self.cond_stage_key]
        if N is not None:
            x, c = x[:N], c[:N]
        return x, c

    def _fine_prefix(self, coarse_position):
        return coarse_position[:, :1]

    def avoid_repeat_or_enforce_pad_for_fine_position(self, logits, sampled_position, flag):
        forbid = torch.zeros_like(logits, dtype=torch.bool)
        forbid.scatter_(1, sampled_position, True)
        forbid[:, self.fine_position_pad_code] = True
        out = _mask_rows(logits, flag, forbid, self.fine_position_eos_code, self.fine_position_pad_code)
        live = ~flag.bool().view(-1)
        out[live, self.fine_position_eos_code + 1:] = -float("Inf")      # the class-label ids sit above <eos>
        return out

    def avoid_special_or_enforce_pad_for_content(self, logits, flag):
        forbid = torch.zeros_like(logits, dtype=torch.bool)
        forbid[:, self.content_pad_code] = True
        forbid[:, self.content_eos_code:] = True                          # <eos> and every class-label id
        return _mask_rows(logits, flag, forbid, None, self.content_pad_code)

    def _fused_rule(self, kind):
        if kind == "fine_pos":       # every id above <eos> (the class labels) is masked on live rows; <eos> keeps its logit
            return dict(pad_code=self.fine_position_pad_code, forbid_codes=(self.fine_position_pad_code,),
                        forbid_from=self.fine_position_eos_code + 1, keep_code=self.fine_position_eos_code)
        if kind == "content":
            return dict(pad_code=self.content_pad_code, forbid_codes=(self.content_pad_code,), forbid_from=self.content_eos_code)
        return super()._fused_rule(kind)
