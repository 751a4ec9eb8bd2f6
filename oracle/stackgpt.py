"""Oracle: StackGPT teacher-forced forward + losses (torch-CPU fp32, functional over a state_dict, autograd-differentiable).
TEST INFRASTRUCTURE -- never imported by the product package.

Follows (behaviour, not code) /root/reference/modules/dynamic_modules/stackgpt.py:44-96 (attention / block) and :175-232
(StackGPT.forward).  Dropout probabilities are taken as 0 (device RNG: "parity unpinned").
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def block(sd, p, x, n_head):
    b, t, c = x.shape
    h = F.layer_norm(x, (c,), sd[p + ".ln1.weight"], sd[p + ".ln1.bias"])
    q = F.linear(h, sd[p + ".attn.query.weight"], sd[p + ".attn.query.bias"]).view(b, t, n_head, c // n_head).transpose(1, 2)
    k = F.linear(h, sd[p + ".attn.key.weight"], sd[p + ".attn.key.bias"]).view(b, t, n_head, c // n_head).transpose(1, 2)
    v = F.linear(h, sd[p + ".attn.value.weight"], sd[p + ".attn.value.bias"]).view(b, t, n_head, c // n_head).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(k.size(-1)))
    mask = torch.tril(torch.ones(t, t, dtype=torch.bool))
    att = att.masked_fill(~mask, float("-inf")).softmax(dim=-1)
    y = (att @ v).transpose(1, 2).contiguous().view(b, t, c)
    x = x + F.linear(y, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    h = F.layer_norm(x, (c,), sd[p + ".ln2.weight"], sd[p + ".ln2.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"])), sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"])
    return x + h


def _stack(sd, name, x, n_head):
    i = 0
    while f"{name}.{i}.ln1.weight" in sd:
        x = block(sd, f"{name}.{i}", x, n_head)
        i += 1
    return x


def _head(sd, name, x):
    c = x.shape[-1]
    return F.linear(F.layer_norm(x, (c,), sd[name + ".0.weight"], sd[name + ".0.bias"]), sd[name + ".1.weight"])


def forward(sd, n_head, coarse_content, fine_content, coarse_position, fine_position, coarse_seg, fine_seg, content_target=None,
            coarse_position_target=None, fine_position_target=None, content_pad=1024, cpos_pad=256, fpos_pad=1024):
    lc = coarse_position.size(1)
    content = torch.cat([coarse_content, fine_content], dim=1)
    # nn.Embedding(padding_idx = the pad code) (stackgpt.py:141-144): same lookup, but the pad row receives NO gradient
    def table(name, pad):
        w = sd[name]
        return lambda idx: F.embedding(idx, w, padding_idx=pad if pad is not None and 0 <= pad < w.shape[0] else None)
    emb_c, emb_cp, emb_fp = table("content_emb.weight", content_pad), table("content_coarse_pos_emb.weight", cpos_pad), table(
        "content_fine_pos_emb.weight", fpos_pad)
    x = emb_c(content[:, :-1])
    pos = torch.cat([emb_cp(coarse_position), emb_fp(fine_position[:, :-1])], dim=1)
    t = pos.shape[1]
    x = x + pos + sd["pos_emb"][:, :t, :]
    if "seg_emb.weight" in sd:
        x = x + sd["seg_emb.weight"][torch.cat([coarse_seg, fine_seg], dim=1)[:, :-1]]
    ph = _stack(sd, "position_transformer", x, n_head)
    upd = torch.cat([emb_cp(coarse_position[:, 1:]), emb_fp(fine_position)], dim=1)
    ch = _stack(sd, "content_transformer", ph + upd, n_head)
    cl, pl = _head(sd, "content_head", ch), _head(sd, "position_head", ph)
    if content_target is None:
        return {"position_logits": pl, "content_logits": cl}
    cpl, fpl = pl[:, :lc - 1], pl[:, lc - 1:]
    coarse = F.cross_entropy(cpl.reshape(-1, cpl.size(-1)), coarse_position_target.reshape(-1), ignore_index=cpos_pad)
    fine = F.cross_entropy(fpl.reshape(-1, fpl.size(-1)), fine_position_target.reshape(-1), ignore_index=fpos_pad)
    con = F.cross_entropy(cl.reshape(-1, cl.size(-1)), content_target.reshape(-1), ignore_index=content_pad)
    return {"position_loss": (coarse + fine) / 2, "content_loss": con, "coarse_position_loss": coarse, "fine_position_loss": fine}
