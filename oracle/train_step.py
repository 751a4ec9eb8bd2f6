"""Oracle: the reference's complete stage-1 training step on CPU (torch autograd over oracle.dqvae / oracle.losses, numpy for
the VQ search, the EMA codebook update and Adam).  TEST INFRASTRUCTURE -- never imported by the product: only tests/,
tools/gen_golden.py (the pin), __graft_entry__.smoke() and bench.py's `cpu_baseline` child process use it.

PINNED: tools/gen_golden.py::gen_train_step drives the REAL reference (DualGrainVQModel + VQLPIPSWithDiscriminator, both
torch.optim.Adam, both LambdaLR schedules) for several steps and aborts unless `run_steps` below reproduces every recorded
quantity; tests/test_oracle_golden.py::test_oracle_train_step_matches_reference repeats the check against the committed fixture
(tests/golden/train_step_*.npz).

What one step is (all under /root/reference):
  * Lightning's automatic optimization with two optimizers = per batch, for i in (0, 1): toggle_optimizer(i),
    training_step(batch, idx, i), zero_grad, backward, optimizer_i.step(); then both `interval: step` schedulers step.
  * models/stage1_dynamic/dqvae_dual_entropy.py:154-183  training_step: a FULL autoencoder forward for each optimizer index --
    in train mode, so the VQ codebook takes TWO EMA updates per batch (the second one sees the autoencoder optimizer 0 just changed);
    :206-231 configure_optimizers: Adam(lr, betas=(0.5, 0.9)) over encoder + decoder + quantize + quant_conv + post_quant_conv
    and over the discriminator; LambdaLR with models/stage1/utils.py:6-26's warm-up / cosine multipliers.
  * modules/vector_quantization/quantize2_mask.py:66-126  search with the OLD weight, EMA buffers, restart of codes whose EMA
    count fell below 1 with rows `vectors[randperm][:K]` (the permutation is an INPUT here), gather with the OLD weight,
    then the weight rewrite.
  * modules/losses/vqperceptual_multidisc.py:109-194  generator branch (L1 + LPIPS + adaptive hinge-GAN weight + codebook loss)
    and discriminator branch (hinge); PatchGAN BatchNorm is in training mode in all three discriminator passes of a step.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import dqvae as odq
from . import entropy as oent
from . import losses as olo
from . import vq as ovq

CB = "quantize.codebook."


# ---- schedule + optimizer (restated; torch.optim is NOT used) -------------------------------------------------------------
def lr_multiplier(scheduler_type, warmup_steps, max_steps, multipler_min, step):
    """models/stage1/utils.py:6-26; LambdaLR evaluates it at step = number of scheduler.step() calls so far"""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    if scheduler_type == "linear-warmup":
        return 1.0
    m = 0.5 * (math.cos((step - warmup_steps) / (max_steps - warmup_steps) * math.pi) + 1)
    return max(m, multipler_min)


def adam_update(p, g, m, v, t, lr, b1, b2, eps=1e-8, weight_decay=0.0, decoupled=True):
    """torch.optim.Adam / AdamW, single tensor, in place on fp32 numpy arrays; t = 1-based step count.
    AdamW (decoupled=True): p *= 1 - lr*wd before the update; Adam with weight_decay: g += wd*p (unused by the reference)."""
    if weight_decay and decoupled:
        p *= np.float32(1.0 - lr * weight_decay)
    elif weight_decay:
        g = g + np.float32(weight_decay) * p
    m += (g - m) * np.float32(1.0 - b1)                      # exp_avg.lerp_(grad, 1 - beta1)
    v *= np.float32(b2)
    v += np.float32(1.0 - b2) * g * g                        # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
    denom = np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)
    p -= np.float32(lr / bc1) * (m / denom)


class Adam:
    """the per-parameter state torch.optim.Adam keeps, over named leaf tensors (updated through .data, like the optimizer)"""

    def __init__(self, named, lr, betas, weight_decay=0.0):
        self.named, self.lr, self.betas, self.wd = dict(named), lr, betas, weight_decay
        self.state = {}

    def step(self, lr):
        for n, p in self.named.items():
            if p.grad is None:                # torch skips parameters without a gradient (the EMA codebook)
                continue
            st = self.state.setdefault(n, {"t": 0, "m": np.zeros(tuple(p.shape), np.float32), "v": np.zeros(tuple(p.shape), np.float32)})
            st["t"] += 1
            pn = p.detach().numpy()           # shares memory with the leaf
            adam_update(pn, p.grad.numpy(), st["m"], st["v"], st["t"], lr, self.betas[0], self.betas[1], weight_decay=self.wd)


# ---- the autoencoder forward in TRAINING mode -------------------------------------------------------------------------------
def ae_forward(sd, x, threshold, beta=0.25, decay=0.99, train=True, restart_perm=None, record=None):
    """(rec, qloss): DualGrainVQModel.forward in train mode.  `sd` holds the whole autoencoder state under the reference's
    keys; with train=True the three VQ entries (cluster_size_ema, embed_ema, codebook weight rows :K) are REWRITTEN in place
    after the gather, as quantize2_mask.py:117-128 orders it.  restart_perm: the permutation standing in for torch.randperm(N)
    (None = restart disabled).  record: optional dict receiving codes / gap of this forward."""
    ent = oent.patch_entropy(x.numpy())
    enc = odq.encoder_dual(sd, x, ent, threshold)
    h = odq.conv(sd, "quant_conv", enc["h_dual"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d)
    w = sd[CB + "weight"]
    cb = w[:-1].detach().clone()
    flat_np = flat.detach().numpy()
    idx_np, gap = ovq.argmin_exact(flat_np, cb.numpy(), return_gap=True)
    idx = torch.from_numpy(idx_np)
    xq = cb[idx]                                           # the OLD weight
    m = enc["codebook_mask"].permute(0, 2, 3, 1).reshape(-1, 1)
    qloss = beta * torch.mean((xq - flat) ** 2 * m) + torch.mean((xq - flat.detach()) ** 2 * m)
    if train:
        k = cb.shape[0]
        rows = None
        if restart_perm is not None:
            assert flat_np.shape[0] >= k, "the tiled-with-noise restart (quantize2_mask.py:57-64) draws rand_like noise: not restated"
            rows = flat_np[np.asarray(restart_perm)][:k]
        n_ema, s_ema, w_new = ovq.ema_update(flat_np, idx_np, sd[CB + "cluster_size_ema"].numpy(), sd[CB + "embed_ema"].numpy(),
                                             decay=decay, restart_rows=rows)
        with torch.no_grad():
            sd[CB + "cluster_size_ema"].copy_(torch.from_numpy(n_ema))
            sd[CB + "embed_ema"].copy_(torch.from_numpy(s_ema))
            w[:-1].copy_(torch.from_numpy(w_new))
    if record is not None:
        record["codes"], record["gap"] = idx_np, gap
    st = flat + (xq - flat).detach()
    z = odq.conv(sd, "post_quant_conv", st.reshape(b, hh, ww, d).permute(0, 3, 1, 2))
    return odq.decoder(sd, z), qloss


def ae_forward_routed(sd, x, n_heads, exponential, beta=0.25, decay=0.99, restart_perm=None, record=None):
    """(rec, qloss, gate, indices): the feature-routed (Gumbel straight-through) DQ-VAE in train mode -- oracle.routing's encoder,
    then the same VQ / EMA / decoder sequence as ae_forward (dqvae_triple_feat.py:83-100, EncoderTriple.py, RouterTriple.py)"""
    from . import routing as oro
    enc = oro.encoder_feature_routed(sd, x, n_heads, exponential)
    h = odq.conv(sd, "quant_conv", enc["h"])
    b, d, hh, ww = h.shape
    flat = h.permute(0, 2, 3, 1).reshape(-1, d)
    w = sd[CB + "weight"]
    cb = w[:-1].detach().clone()
    flat_np = flat.detach().numpy()
    idx_np, gap = ovq.argmin_exact(flat_np, cb.numpy(), return_gap=True)
    xq = cb[torch.from_numpy(idx_np)]
    m = enc["codebook_mask"].permute(0, 2, 3, 1).reshape(-1, 1)
    qloss = beta * torch.mean((xq - flat) ** 2 * m) + torch.mean((xq - flat.detach()) ** 2 * m)
    k = cb.shape[0]
    rows = flat_np[np.asarray(restart_perm)][:k] if restart_perm is not None else None
    n_ema, s_ema, w_new = ovq.ema_update(flat_np, idx_np, sd[CB + "cluster_size_ema"].numpy(), sd[CB + "embed_ema"].numpy(), decay=decay,
                                         restart_rows=rows)
    with torch.no_grad():
        sd[CB + "cluster_size_ema"].copy_(torch.from_numpy(n_ema))
        sd[CB + "embed_ema"].copy_(torch.from_numpy(s_ema))
        w[:-1].copy_(torch.from_numpy(w_new))
    if record is not None:
        record["codes"], record["gap"], record["indices"] = idx_np, gap, enc["indices"].numpy()
    st = flat + (xq - flat).detach()
    z = odq.conv(sd, "post_quant_conv", st.reshape(b, hh, ww, d).permute(0, 3, 1, 2))
    return odq.decoder(sd, z), qloss, enc["gate"]


def budget_triple(gate, target_fine_ratio, target_median_ratio, gamma, min_grain_size, median_grain_size, max_grain_size):
    """BudgetConstraint_NormedSeperateRatioMSE_TripleGrain (modules/dynamic_modules/budget.py:30-60); gate [B, 3, h, w]"""
    min_c = min_grain_size * min_grain_size
    med_c = median_grain_size * median_grain_size - min_c
    max_c = max_grain_size * max_grain_size - min_c
    b = gate.shape[0]
    r_med = ((gate[:, 0] + 4.0 * gate[:, 1] + gate[:, 2]).sum() / b - min_c) / med_c
    r_fine = ((gate[:, 0] + 16.0 * gate[:, 2] + gate[:, 1]).sum() / b - min_c) / max_c
    return gamma * (r_fine - target_fine_ratio) ** 2 + (r_med - target_median_ratio) ** 2


def _split_state(state):
    sd = {k: v for k, v in state.items() if not k.startswith("loss.")}
    sd_d = {k[len("loss.discriminator."):]: v for k, v in state.items() if k.startswith("loss.discriminator.")}
    sd_l = {k[len("loss.perceptual_loss."):]: v for k, v in state.items() if k.startswith("loss.perceptual_loss.")}
    return sd, sd_d, sd_l


def run_steps(state, param_keys, batches, threshold, lr, min_lr=0.0, warmup_steps=0, max_steps=1,
              scheduler_type="linear-warmup_cosine-decay", restart_perm=None, disc_weight_max=0.75, perceptual_weight=1.0,
              disc_factor=1.0, watch=(), stride=lambda n: 1, record_grads=True, routed=None):
    """The reference's two-optimizer schedule for len(batches) steps, in place on `state` ({reference state_dict key: fp32 torch
    tensor}, autoencoder + `loss.discriminator.*` + `loss.perceptual_loss.*`; buffers included).  param_keys: the names that are
    nn.Parameters (the rest are buffers).  Returns {fixture key: value} with the keys tools/gen_golden.py::run_reference_train_steps
    writes (s<step>.o<i>.{lr,loss,codes,gap,cluster_size_ema,embed_ema,codebook}, s<step>.log.*, s<step>.{param,exp_avg,exp_avg_sq}.<name>,
    s0.grad.<name>, final.disc_buf.*)."""
    sd, sd_d, sd_l = _split_state(state)
    pk = set(str(k) for k in param_keys)
    ae_named = {k: state[k] for k in state if k in pk and not k.startswith("loss.")}
    d_named = {k: state[k] for k in state if k in pk and k.startswith("loss.discriminator.")}
    for k, v in ae_named.items():
        v.requires_grad_(k != CB + "weight")           # the EMA codebook is a frozen parameter inside optimizer 0
    for v in d_named.values():
        v.requires_grad_(False)
    opt_ae, opt_d = Adam(ae_named, lr, (0.5, 0.9)), Adam(d_named, lr, (0.5, 0.9))
    mult_min = (min_lr / lr) if lr else 0.0
    sched_t = 0
    out = {}

    def sample(a):
        a = np.asarray(a).reshape(-1)
        return a[:: stride(a.size)].astype(np.float32).copy()

    def snap(pre):
        out[pre + "cluster_size_ema"] = sd[CB + "cluster_size_ema"].numpy().copy()
        kk = sd[CB + "embed_ema"].shape[0]
        out[pre + "embed_ema"] = sd[CB + "embed_ema"].numpy()[:: max(1, kk // 64)].copy()          # 64 rows of K
        out[pre + "codebook"] = sd[CB + "weight"].detach().numpy()[:-1][:: max(1, kk // 64)].copy()

    def set_grad(named, on):
        for k, v in named.items():
            v.requires_grad_(on and k != CB + "weight")
            v.grad = None

    for step, x in enumerate(batches):
        x = torch.as_tensor(x)
        perm = restart_perm[step] if isinstance(restart_perm, (list, tuple)) else restart_perm
        cur_lr = lr * lr_multiplier(scheduler_type, warmup_steps, max_steps, mult_min, sched_t)
        # ---- optimizer 0: the autoencoder (discriminator parameters frozen by toggle_optimizer) ----
        set_grad(ae_named, True)
        set_grad(d_named, False)
        rec_info = {}
        bl = None
        if routed is None:
            rec, qloss = ae_forward(sd, x, threshold, restart_perm=perm, record=rec_info)
        else:
            rec, qloss, gate = ae_forward_routed(sd, x, routed["n_heads"], torch.as_tensor(routed["exponential"][step]), restart_perm=perm,
                                                 record=rec_info)
            bl = budget_triple(gate, **routed["budget"])
        run = {}
        r = olo.generator_loss(sd_d, sd_l, x, rec, qloss, sd["decoder.conv_out.weight"], perceptual_weight=perceptual_weight,
                               disc_factor=disc_factor, disc_weight_max=disc_weight_max, running=run)
        if bl is not None:
            r["loss"] = r["loss"] + bl
        r["loss"].backward()
        _bn_commit(sd_d, run, 1)
        if step == 0 and record_grads:
            for n in watch:
                if n in ae_named and ae_named[n].grad is not None:
                    out[f"s0.grad.{n}"] = sample(ae_named[n].grad.numpy())
        opt_ae.step(cur_lr)
        pre = f"s{step}.o0."
        out[pre + "lr"], out[pre + "loss"] = np.float64(cur_lr), np.float32(r["loss"].item())
        out[pre + "codes"], out[pre + "gap"] = rec_info["codes"].astype(np.int16), rec_info["gap"].astype(np.float32)
        snap(pre)
        lg = f"s{step}.log."
        out[lg + "train_aeloss"] = np.float32(r["loss"].item())
        out[lg + "train_total_loss"] = np.float32(r["loss"].item())
        out[lg + "train_quant_loss"] = np.float32(qloss.item())
        out[lg + "train_nll_loss"] = np.float32(r["nll"].item())
        out[lg + "train_rec_loss"] = np.float32(r["rec_mean"].item())
        out[lg + "train_p_loss"] = np.float32(r["p"].mean().item())
        out[lg + "train_d_weight"] = np.float32(r["d_weight"].item())
        out[lg + "train_disc_factor"] = np.float32(disc_factor)
        out[lg + "train_g_loss"] = np.float32(r["g"].item())
        if routed is None:
            out[lg + "train_fine_ratio"] = np.float32(oent.entropy_gate(oent.patch_entropy(x.numpy()), threshold)[..., 1].mean())
        else:
            out[lg + "train_budget_loss"] = np.float32(bl.item())
            out[lg + "train_fine_radio"] = np.float32((rec_info["indices"] == 2).mean())
            out[lg + "train_median_radio"] = np.float32((rec_info["indices"] == 1).mean())
        # ---- optimizer 1: the discriminator, on a SECOND training-mode autoencoder forward ----
        set_grad(ae_named, False)
        set_grad(d_named, True)
        rec_info = {}
        with torch.no_grad():
            if routed is None:
                rec2, _ = ae_forward(sd, x, threshold, restart_perm=perm, record=rec_info)
            else:
                rec2, _, _ = ae_forward_routed(sd, x, routed["n_heads"], torch.as_tensor(routed["exponential"][step]), restart_perm=perm,
                                               record=rec_info)
        run = {}
        d_loss, lr_, lf_ = olo.discriminator_loss(sd_d, x, rec2, disc_factor=disc_factor, running=run)
        d_loss.backward()
        _bn_commit(sd_d, run, 2)
        if step == 0 and record_grads:
            for n in watch:
                if n in d_named and d_named[n].grad is not None:
                    out[f"s0.grad.{n}"] = sample(d_named[n].grad.numpy())
        opt_d.step(cur_lr)
        pre = f"s{step}.o1."
        out[pre + "lr"], out[pre + "loss"] = np.float64(cur_lr), np.float32(d_loss.item())
        out[pre + "codes"], out[pre + "gap"] = rec_info["codes"].astype(np.int16), rec_info["gap"].astype(np.float32)
        snap(pre)
        out[lg + "train_discloss"] = np.float32(d_loss.item())
        out[lg + "train_disc_loss"] = np.float32(d_loss.item())
        out[lg + "train_logits_real"] = np.float32(lr_.mean().item())
        out[lg + "train_logits_fake"] = np.float32(lf_.mean().item())
        sched_t += 1
        for n in watch:
            opt = opt_d if n in d_named else opt_ae
            out[f"s{step}.param.{n}"] = sample(state[n].detach().numpy())
            out[f"s{step}.exp_avg.{n}"] = sample(opt.state[n]["m"])
            out[f"s{step}.exp_avg_sq.{n}"] = sample(opt.state[n]["v"])
    set_grad(ae_named, False)
    set_grad(d_named, False)
    for k, v in sd_d.items():
        if k not in {n[len("loss.discriminator."):] for n in d_named}:
            out["final.disc_buf." + k] = v.numpy().copy()
    return out


def _bn_commit(sd_d, run, n_calls):
    """write the BatchNorm running statistics a discriminator pass left (oracle.losses.patchgan's `running`) back into the state and
    count the passes (num_batches_tracked)"""
    with torch.no_grad():
        for k, v in run.items():
            sd_d[k].copy_(v)
        for k in sd_d:
            if k.endswith("num_batches_tracked"):
                sd_d[k] += n_calls


def _l2(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b)) / max(1e-30, float(np.linalg.norm(b)))


def compare_records(got, ref, start_param=None, skip=()):
    """Distances between two records of the pinned run (run_steps' / the fixture's keys) -> {key: (kind, err)}.

    kind / err:
      "codes"  rows whose code differs AND whose recorded fp64 top-2 gap (the reference's own VQ input) is >= 1e-4: must be 0;
               "codes_near" = the differing rows with a smaller gap (rounding-level near ties; reported, bounded by the caller)
      "scalar" |a - b| / max(|b|, 1e-3)                      losses, logged terms, lr
      "l2"     ||a - b|| / ||b||                              gradients, Adam moments, EMA buffers, codebook, BatchNorm statistics
      "dparam" ||(a - p0) - (b - p0)|| / ||b - p0|| with p0 = start_param(name) (the sampled start value): the parameter MOVEMENT;
               when the reference did not move (the lr = 0 warm-up step) err = max|a - p0| and must be exactly 0
    Why L2 and not max-norm: the LPIPS / PatchGAN gradients flip isolated ReLU / LeakyReLU / max-pool decisions on rounding-level
    pre-activations, which changes single gradient entries by O(1) of their size in ANY fp32 implementation (measured while building
    the fixture: the reference's LPIPS / GAN parameter gradients differ from an fp64 evaluation of the same graph by 0.4 - 0.5 % of
    the tensor's max, its L1 / codebook-loss gradients by 1.5e-5)."""
    res = {}
    # "late" forwards = those that run after the first autoencoder update with lr > 0: from there on the two runs no longer start from
    # bit-identical parameters (Adam's sign-level differences), so code flips at small gaps and their EMA consequences are expected
    first_real = min([int(k.split(".")[0][1:]) for k in ref if k.endswith(".o0.lr") and float(ref[k]) > 0] or [1 << 30])

    def late(key):
        parts = key.split(".")
        if not (parts[0][0] == "s" and parts[0][1:].isdigit() and len(parts) > 1 and parts[1] in ("o0", "o1")):
            return False
        st, oi = int(parts[0][1:]), int(parts[1][1:])
        return (st, oi) > (first_real, 0)

    for key in ref:
        if key in skip or key not in got or key.startswith(("state_", "param_keys", "decay_names")) or key.endswith(".gap"):
            continue
        a, b = np.asarray(got[key]), np.asarray(ref[key])
        sfx = "_late" if late(key) else ""
        if key.endswith(".codes"):
            gap = np.asarray(ref[key[:-len("codes")] + "gap"])
            bad = a.reshape(-1) != b.reshape(-1)
            res[key] = ("codes" + sfx, int((bad & (gap >= 1e-4)).sum()))
            res[key + "_near"] = ("codes_near" + sfx, int((bad & (gap < 1e-4)).sum()))
        elif key.endswith("num_batches_tracked"):
            res[key] = ("scalar", float(abs(int(a) - int(b))))
        elif b.ndim == 0:
            res[key] = ("scalar", abs(float(a) - float(b)) / max(abs(float(b)), 1e-3))
        elif ".param." in key and start_param is not None:
            p0 = start_param(key.split(".param.")[1])
            db = b.astype(np.float64) - p0
            da = a.astype(np.float64) - p0
            if np.abs(db).max() == 0:
                res[key] = ("dparam0", float(np.abs(da).max()))
            else:
                res[key] = ("dparam", float(np.linalg.norm(da - db) / np.linalg.norm(db)))
        else:
            res[key] = ("l2" + sfx, _l2(a, b))
    return res


def summarize(cmp):
    """{(step, group): worst err} of a compare_records result, for printing / bounding by group"""
    out = {}
    for key, (kind, err) in cmp.items():
        parts = key.split(".")
        step = parts[0]
        disc = "_disc" if ".loss.discriminator." in key else ""          # the discriminator's tensors: see PIN_BOUNDS
        if kind in ("dparam", "dparam0"):
            grp = kind + disc
        elif ".exp_avg_sq." in key:
            grp = "exp_avg_sq" + disc
        elif ".exp_avg." in key:
            grp = "exp_avg" + disc
        elif ".grad." in key:
            grp = "grad" + disc
        elif kind.startswith("codes"):
            grp = kind
        elif step == "final":
            grp = "disc_buf"
        elif kind == "scalar":
            grp = "scalar:" + parts[-1]
        else:
            grp = parts[-1] + ("_late" if kind.endswith("_late") else "")
        out[(step, grp)] = max(out.get((step, grp), 0.0), err)
    return out


# worst allowed distance per summarize() group.  "cpu" = this oracle against the reference (both fp32 on the host: summation order
# only); the GPU levels are set in tests/test_gpu_trainstep.py from measurements on MI355X.
PIN_BOUNDS = {
    "codes": 0, "codes_near": 1, "codes_late": 8, "codes_near_late": 12,
    "cluster_size_ema": 1e-5, "embed_ema": 1e-5, "codebook": 1e-5,
    "cluster_size_ema_late": 5e-3, "embed_ema_late": 5e-3, "codebook_late": 5e-3,
    "scalar:lr": 1e-12, "scalar:loss": 2e-4, "scalar": 5e-3, "s2:scalar:loss": 2e-3, "s2:scalar": 3e-2,
    # means of signed PatchGAN logits near zero (|mean| ~ 0.1 of values ~ +-1): cancellation
    "scalar:train_logits_fake": 5e-2, "scalar:train_logits_real": 5e-2, "s2:scalar:train_logits_fake": 5e-2, "s2:scalar:train_logits_real": 5e-2,
    "grad": 5e-2, "exp_avg": 5e-2, "exp_avg_sq": 5e-2, "s2:exp_avg": 0.15, "s2:exp_avg_sq": 0.15,
    # the discriminator's hinge gradient is 0.5 * (grad mean D(rec) - grad mean D(x)): a small difference of two large, nearly equal
    # terms while D cannot tell the two apart, so rounding in D's products is amplified ~1e4 x in ITS parameter gradients
    "grad_disc": 5e-2, "exp_avg_disc": 5e-2, "exp_avg_sq_disc": 5e-2, "s2:exp_avg_disc": 0.15, "s2:exp_avg_sq_disc": 0.15,
    "dparam0": 0.0, "dparam": 0.2, "dparam0_disc": 0.0, "dparam_disc": 0.2, "disc_buf": 5e-3,
}


def check_summary(summary, bounds=PIN_BOUNDS):
    """[(step, group, err, bound)] of the groups that exceed their bound"""
    bad = []
    for (step, grp), err in sorted(summary.items()):
        b = bounds.get(f"{step}:{grp}", bounds.get(grp, bounds["scalar"] if grp.startswith("scalar:") else None))     # "s2:codes_near" overrides "codes_near"
        assert b is not None, grp
        if err > b:
            bad.append((step, grp, err, b))
    return bad


def sampled_start_param(meta, k, zc, stride, scale=None):
    """name -> the strided sample of the pinned start value of a parameter (fp64), for compare_records(start_param=...)"""
    from dynamicvectorquantization_amd import synth
    shapes = {str(kk): tuple(int(v) for v in str(s).split(",")) if str(s) else () for kk, s in zip(meta["state_keys"], meta["state_shapes"])}

    def p0(name):
        a = synth.train_step_param(name, shapes[name], k, zc).reshape(-1)
        if scale and name in scale:
            a = (a * np.float32(scale[name])).astype(np.float32)
        return a[:: stride(a.size)].astype(np.float64)
    return p0


def reference_schedule_steps(tag, meta):
    """the pinned run of tests/golden_cfg.TRAIN_STEP[tag] from its deterministic start state; meta = the fixture's
    state_keys / state_shapes / param_keys"""
    import os
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "tests"))
    from golden_cfg import TRAIN_STEP, TRAIN_STEP_WATCH, train_step_stride
    from dynamicvectorquantization_amd import synth
    c = TRAIN_STEP[tag]
    g = synth.DQVAE_GEOM[c["geom"]]
    state = pinned_start_state(meta, g["k"], g["zc"])
    thr = oent.threshold_from_table(os.path.join(here, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    perms = [synth.train_step_restart_perm(s_, c["bs"], g["k"], g["resolution"]) for s_ in range(c["steps"])]
    return run_steps(state, [str(k) for k in meta["param_keys"]], synth.train_step_batches(c["steps"], c["bs"], g["resolution"]), thr,
                     lr=c["lr"], min_lr=c["min_lr"], warmup_steps=c["steps_per_epoch"] * c["warmup_epochs"], max_steps=c["training_steps"],
                     restart_perm=perms, watch=TRAIN_STEP_WATCH, stride=train_step_stride)


def triple_schedule_steps(meta):
    """the pinned triple-grain run (tests/golden_cfg.TRAIN_STEP_TRIPLE); meta = the fixture (state_keys / state_shapes / param_keys and the
    per-step restart permutations `s<step>.perm`, which were derived from the reference's own grain maps)"""
    import os
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "tests"))
    from golden_cfg import TRAIN_STEP_TRIPLE, TRAIN_STEP_TRIPLE_WATCH, TRIPLE_BUDGET, train_step_stride
    from dynamicvectorquantization_amd import synth
    c = TRAIN_STEP_TRIPLE
    state = pinned_start_state(meta, c["k"], c["zc"], scale={"encoder.router.gate.2.weight": c["last_gate_scale"]})
    perms = [np.asarray(meta[f"s{s_}.perm"]) for s_ in range(c["steps"])]
    routed = dict(n_heads=3, exponential=[synth.train_step_gumbel(s_, c["bs"]) for s_ in range(c["steps"])], budget=dict(TRIPLE_BUDGET))
    return run_steps(state, [str(k) for k in meta["param_keys"]], synth.train_step_batches(c["steps"], c["bs"], 64), None,
                     lr=c["lr"], min_lr=c["min_lr"], warmup_steps=c["steps_per_epoch"] * c["warmup_epochs"], max_steps=c["training_steps"],
                     restart_perm=perms, watch=TRAIN_STEP_TRIPLE_WATCH, stride=train_step_stride, routed=routed)


def pinned_start_state(meta, k, zc, scale=None):
    """{key: tensor} of the pinned run's start: parameters from synth.train_step_param, VQ EMA buffers from synth.train_step_vq_state,
    BatchNorm buffers at their constructor values, LPIPS ScalingLayer constants"""
    from dynamicvectorquantization_amd import synth
    pk = set(str(s) for s in meta["param_keys"])
    state = {}
    for key, shp in zip(meta["state_keys"], meta["state_shapes"]):
        key = str(key)
        shape = tuple(int(v) for v in str(shp).split(",")) if str(shp) else ()
        if key in pk:
            v = synth.train_step_param(key, shape, k, zc).copy()
            if scale and key in scale:
                v = (v * np.float32(scale[key])).astype(np.float32)
            state[key] = torch.from_numpy(v)
        elif key.endswith("running_var"):
            state[key] = torch.ones(shape)
        elif key.endswith("num_batches_tracked"):
            state[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith("scaling_layer.shift"):
            state[key] = torch.tensor(olo.LPIPS_SHIFT).reshape(shape)
        elif key.endswith("scaling_layer.scale"):
            state[key] = torch.tensor(olo.LPIPS_SCALE).reshape(shape)
        else:
            state[key] = torch.zeros(shape)
    n0, s0 = synth.train_step_vq_state(k, zc)
    state[CB + "cluster_size_ema"] = torch.from_numpy(n0.copy())
    state[CB + "embed_ema"] = torch.from_numpy(s0.copy())
    return state


# ---- stage 2: Dualformer + AdamW ---------------------------------------------------------------------------------------------------
PIN_BOUNDS_S2 = {"scalar:lr": 1e-12, "scalar:loss": 1e-5, "scalar": 1e-4, "grad": 1e-4, "exp_avg": 1e-4, "exp_avg_sq": 2e-4,
                 "dparam0": 0.0, "dparam": 2e-3}


def decay_groups(names):
    """dqtransformer_uncond_entropy.py:92-125 by parameter NAME over StackGPT's module tree: biases, LayerNorm (ln*, the heads' first
    element) and Embedding (`*_emb`) weights and `pos_emb` are not decayed; every other `.weight` belongs to an nn.Linear and is"""
    decay, no_decay = set(), set()
    for n in names:
        leaf = n.split(".")
        if n == "pos_emb" or leaf[-1] == "bias":
            no_decay.add(n)
        elif leaf[-1] == "weight" and (leaf[0].endswith("_emb") or leaf[-2].startswith("ln") or (leaf[0].endswith("_head") and leaf[1] == "0")):
            no_decay.add(n)
        else:
            decay.add(n)
    return decay, no_decay


def _dualformer_start(meta):
    import os
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "tests"))
    from dynamicvectorquantization_amd import synth
    shapes = {str(k): tuple(int(v) for v in str(s).split(",")) if str(s) else () for k, s in zip(meta["state_keys"], meta["state_shapes"])}
    sd_gpt = {}
    for k, shp in shapes.items():
        if not k.startswith("transformer.") or k.endswith("attn.mask"):
            continue
        n = k[len("transformer."):]
        sd_gpt[n] = torch.from_numpy(synth.det_param("dualformer.uncond." + n, shp) * np.float32(0.3 if n == "pos_emb" else 1.0))
    return sd_gpt, shapes, here


def dualformer_start_param(meta):
    from golden_cfg import train_step_stride
    sd_gpt, _, _ = _dualformer_start(meta)

    def p0(name):
        a = sd_gpt[name].numpy().reshape(-1)
        return a[:: train_step_stride(a.size)].astype(np.float64)
    return p0


def dualformer_schedule_steps(meta):
    """the pinned stage-2 run (tests/golden_cfg.TRAIN_STEP_S2): teacher-forced forward (oracle.dualformer), total loss =
    content + 0.7 * position (golden_cfg.dualformer_cfg), AdamW(.9, .95) on the two groups, the warm-up / cosine multipliers"""
    from . import dualformer as odf
    sd_gpt, shapes, here = _dualformer_start(meta)
    from conftest import load_golden
    from golden_cfg import TRAIN_STEP_S2, TRAIN_STEP_S2_WATCH, dualformer_cfg, train_step_s2_batch, train_step_stride
    from test_oracle_golden import dqvae_state_dict
    c = TRAIN_STEP_S2
    cfg = dualformer_cfg("uncond")
    sd_first = dqvae_state_dict(load_golden("dqvae_small"), "spread", 512, 64)
    thr = oent.threshold_from_table(os_path_join(here, "scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    decay, no_decay = decay_groups(sd_gpt.keys())
    assert sorted(decay) == [str(n) for n in meta["decay_names"]], sorted(decay ^ set(str(n) for n in meta["decay_names"]))
    for v in sd_gpt.values():
        v.requires_grad_(True)
    opt_decay = Adam({n: sd_gpt[n] for n in sorted(decay)}, c["lr"], (0.9, 0.95), weight_decay=c["weight_decay"])
    opt_plain = Adam({n: sd_gpt[n] for n in sorted(no_decay)}, c["lr"], (0.9, 0.95))
    n_head = cfg["transformer_config"]["params"]["n_head"]
    out = {}

    def sample(a):
        a = np.asarray(a).reshape(-1)
        return a[:: train_step_stride(a.size)].astype(np.float32).copy()

    for step in range(c["steps"]):
        lr = c["lr"] * lr_multiplier("linear-warmup_cosine-decay", c["steps_per_epoch"] * c["warmup_epochs"], c["training_steps"],
                                     c["min_lr"] / c["lr"], step)
        for v in sd_gpt.values():
            v.grad = None
        o = odf.forward(sd_first, sd_gpt, n_head, torch.from_numpy(train_step_s2_batch(step)), thr, kind="uncond")
        loss = cfg["content_loss_weight"] * o["content_loss"] + cfg["position_loss_weight"] * o["position_loss"]
        loss.backward()
        if step == 0:
            for n in TRAIN_STEP_S2_WATCH:
                out[f"s0.grad.{n}"] = sample(sd_gpt[n].grad.numpy())
        opt_decay.step(lr)
        opt_plain.step(lr)
        out[f"s{step}.lr"], out[f"s{step}.loss"] = np.float64(lr), np.float32(loss.item())
        for kk in ("content_loss", "position_loss", "coarse_position_loss", "fine_position_loss"):
            out[f"s{step}.log.train_{kk}"] = np.float32(float(o[kk].detach()))
        out[f"s{step}.log.train_loss"] = np.float32(loss.item())
        for n in TRAIN_STEP_S2_WATCH:
            st = (opt_decay if n in decay else opt_plain).state[n]
            out[f"s{step}.param.{n}"] = sample(sd_gpt[n].detach().numpy())
            out[f"s{step}.exp_avg.{n}"] = sample(st["m"])
            out[f"s{step}.exp_avg_sq.{n}"] = sample(st["v"])
    return out


def os_path_join(*a):
    import os
    return os.path.join(*a)


# ---- bench.py's cpu_baseline legs (same step, reference-initialised state, no records) ------------------------------------
def full_objective_steps(state, param_keys, batches, threshold, lr=1e-4, steps=1, disc_weight_max=0.75, restart_perm=None):
    """`steps` complete two-optimizer steps (both autoencoder forwards with their EMA updates, LPIPS, adaptive GAN weight, both Adam
    updates) in place on `state`; returns [(aeloss, discloss)]"""
    bs = [batches[s % len(batches)] for s in range(steps)]
    o = run_steps(state, param_keys, bs, threshold, lr=lr, max_steps=max(2, steps), restart_perm=restart_perm,
                  disc_weight_max=disc_weight_max, record_grads=False)
    return [(float(o[f"s{s}.o0.loss"]), float(o[f"s{s}.o1.loss"])) for s in range(steps)]


def ae_only_steps(state, param_keys, batches, threshold, lr=1e-4, steps=1, restart_perm=None):
    """the AE-only objective (perceptual_weight = disc_factor = 0 in the lossconfig: L1 + codebook loss): one training-mode forward
    with its EMA update, backward, Adam -- what the HIP trainer runs under `--objective ae`.  Returns the losses."""
    sd, _, _ = _split_state(state)
    pk = set(str(k) for k in param_keys)
    named = {k: state[k] for k in state if k in pk and not k.startswith("loss.")}
    for k, v in named.items():
        v.requires_grad_(k != CB + "weight")
    opt = Adam(named, lr, (0.5, 0.9))
    losses = []
    for s in range(steps):
        for v in named.values():
            v.grad = None
        x = torch.as_tensor(batches[s % len(batches)])
        rec, qloss = ae_forward(sd, x, threshold, restart_perm=restart_perm)
        loss = torch.mean(torch.abs(x - rec)) + qloss
        loss.backward()
        opt.step(lr)
        losses.append(float(loss))
    return losses
