"""GPU parity of the fused causal attention kernels (csrc/attention.hip: bf16, head size 64 / 128) against an fp32 torch
restatement of CausalSelfAttention.forward (/root/reference/modules/dynamic_modules/stackgpt.py:41-69) and against the
package's own unfused per-head GEMM path; head sizes 64 and 128 (the shipped p6c18 configs use 8 heads of 128).  Tolerance: rel-to-max 2e-2 (bf16 operands, fp32 accumulation)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _reference(q, k, v, dout, b, t, nh, mask=None, hs=64):
    """fp64 autograd restatement; q, k, v, dout numpy [b*t, nh*hs]; mask [b, nh, t, t] multiplicative (dropout) or None"""
    qs, ks, vs = (torch.from_numpy(a).double().view(b, t, nh, hs).transpose(1, 2).requires_grad_(True) for a in (q, k, v))
    att = (qs @ ks.transpose(-2, -1)) * (1.0 / math.sqrt(hs))
    causal = torch.tril(torch.ones(t, t, dtype=torch.bool))
    att = att.masked_fill(~causal, float("-inf")).softmax(dim=-1)
    if mask is not None:
        att = att * torch.from_numpy(mask).double()
    y = (att @ vs).transpose(1, 2).reshape(b * t, nh * hs)
    (y * torch.from_numpy(dout).double()).sum().backward()
    un = lambda g: g.transpose(1, 2).reshape(b * t, nh * hs).numpy()
    return y.detach().numpy(), un(qs.grad), un(ks.grad), un(vs.grad)


def _rel(got, ref):
    return float(np.abs(got - ref).max() / max(1e-9, np.abs(ref).max()))


@pytest.mark.parametrize("shape", [(2, 72, 2), (1, 32, 1), (3, 200, 4), (1, 648, 2)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("p_drop", [0.0, 0.25])
@pytest.mark.parametrize("hs", [64, 128])
def test_fused_attention_vs_fp64_reference(dev, shape, p_drop, hs):
    from dynamicvectorquantization_amd import kernels as K
    b, t, nh = shape
    c = nh * hs
    scale = 1.0 / math.sqrt(hs)
    rs = np.random.RandomState(b * 1000 + t)
    q, k, v, dout = (_bf16(rs.standard_normal((b * t, c)).astype(np.float32) * s) for s in (1.5, 1.5, 1.0, 1.0))
    seed = 0x1234_5678_9ABC + t
    mask = None
    if p_drop > 0:
        # the attention-dropout mask is DEFINED as dvq_dropout's mask over the flat [b, nh, t, t] probability tensor
        ones = torch.ones(b * nh * t * t, dtype=torch.bfloat16, device=dev)
        mask = (K.dropout(ones, p_drop, seed).float() > 0).float().view(b, nh, t, t).cpu().numpy() / (1.0 - p_drop)
        keep = float((mask > 0).mean())
        assert abs(keep - (1 - p_drop)) < 0.02, keep
    yr, dqr, dkr, dvr = _reference(q, k, v, dout, b, t, nh, mask, hs)
    dq_, dk_, dv_, do_ = (torch.from_numpy(a).to(dev, torch.bfloat16) for a in (q, k, v, dout))
    assert K.attn_causal_ok(dq_, nh, b, t)
    y, lse = K.attn_causal_fwd(dq_, dk_, dv_, b, t, nh, scale, p_drop, seed)
    gq, gk, gv = K.attn_causal_bwd(dq_, dk_, dv_, y, do_, lse, b, t, nh, scale, p_drop, seed)
    torch.cuda.synchronize()
    assert _rel(y.float().cpu().numpy(), yr) < 2e-2
    # log-sum-exp of the scaled causal scores (saved for the backward)
    sc = (torch.from_numpy(q).view(b, t, nh, hs).transpose(1, 2) @ torch.from_numpy(k).view(b, t, nh, hs).transpose(1, 2).transpose(-2, -1)) * scale
    sc = sc.masked_fill(~torch.tril(torch.ones(t, t, dtype=torch.bool)), float("-inf"))
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(sc, dim=-1).numpy(), rtol=2e-3, atol=2e-3)
    for name, got, ref in (("dq", gq, dqr), ("dk", gk, dkr), ("dv", gv, dvr)):
        assert _rel(got.float().cpu().numpy(), ref) < 2e-2, name


@pytest.mark.parametrize("shape", [(2, 72, 2), (3, 200, 4), (2, 776, 3)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("hs", [64, 128])
def test_fused_attention_drop_mask_equals_rehash(dev, shape, hs):
    """the keep decisions the forward leaves in the drop-mask buffer (1 bit per score, csrc/attention.hip: drop_tile) give the
    backward kernels exactly the dropout they would hash themselves: outputs and all three gradients are bit-identical"""
    from dynamicvectorquantization_amd import kernels as K
    b, t, nh = shape
    c, scale, p_drop, seed = nh * hs, 1.0 / math.sqrt(hs), 0.1, 0xABCDEF12345 + t
    g = torch.Generator(device="cpu").manual_seed(t + hs)
    q, k, v, do = (torch.randn(b * t, c, generator=g).to(dev, torch.bfloat16) for _ in range(4))
    y0, lse0 = K.attn_causal_fwd(q, k, v, b, t, nh, scale, p_drop, seed)
    ref = K.attn_causal_bwd(q, k, v, y0, do, lse0, b, t, nh, scale, p_drop, seed)
    dm = K.attn_causal_drop_mask(q, b, t, nh)
    dm.fill_(-1)                                             # stale contents must not matter: every causal tile is rewritten
    y1, lse1 = K.attn_causal_fwd(q, k, v, b, t, nh, scale, p_drop, seed, drop_mask=dm)
    got = K.attn_causal_bwd(q, k, v, y1, do, lse1, b, t, nh, scale, p_drop, seed, drop_mask=dm)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(lse0, lse1)
    # same decisions -> same gradients; the two kernel variants are separate compilations of the element-wise code (fp contraction
    # may differ), so "same" is: a vanishing fraction of elements off by at most one bf16 rounding step
    for name, a, r in zip(("dq", "dk", "dv"), got, ref):
        d = (a.float() - r.float()).abs()
        n_off, worst = int((d > 0).sum()), float(d.max() / r.float().abs().max())
        print(f"drop-mask vs rehash {name}: {n_off} of {d.numel()} elements differ, worst {worst:.2e} of max|ref|")
        assert n_off <= 1e-3 * d.numel() and worst < 4e-3, (name, n_off, worst)
    # the buffer really carries the decisions: the kept fraction of the causal tiles' bits is 1 - p
    nt = (t + 31) // 32
    w = dm.view(b * nh, nt, nt, 16)
    tri = torch.tril(torch.ones(nt, nt, dtype=torch.bool, device=dev), diagonal=-1)          # strictly-below-diagonal tiles: all valid
    if int(tri.sum()) > 0 and t % 32 == 0:
        bits = w[:, tri].cpu().numpy().view(np.uint64)
        frac = float(np.unpackbits(bits.view(np.uint8)).mean())
        assert abs(frac - (1 - p_drop)) < 0.02, frac


def test_fused_attention_rejects_unsupported_geometry(dev):
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd._lib import DvqError
    x = torch.zeros(2 * 36, 64, dtype=torch.bfloat16, device=dev)
    assert not K.attn_causal_ok(x, 1, 2, 36)              # T % 8 != 0
    assert not K.attn_causal_ok(x.float(), 1, 2, 36)
    assert not K.attn_causal_ok(x, 2, 2, 40)              # head size 32
    with pytest.raises(DvqError):
        K.attn_causal_fwd(x, x, x, 2, 36, 1, 0.125)


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("n_head", [8, 16])
def test_attention_module_fused_equals_unfused_full_size(dev, train, n_head):
    """CausalSelfAttention of the p6c18 transformer (1024 channels, 8 heads of 128; also 16 x 64; T = 648), fwd + bwd: the fused kernels
    against the per-head GEMM path on identical weights, inputs and dropout seeds"""
    from types import SimpleNamespace
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import stackgpt as sg
    from dynamicvectorquantization_amd.layers import Tape
    cfg = SimpleNamespace(n_embd=1024, n_head=n_head, attn_pdrop=0.1, resid_pdrop=0.1, block_size=648)
    b, t = 2, 648
    torch.manual_seed(5)
    with rt.compute_dtype_ctx(torch.bfloat16):
        attn = sg.CausalSelfAttention(cfg).to(dev)
        attn.train(train)
        x = (torch.randn(b * t, 1024, device=dev) * 1.0).to(torch.bfloat16)
        dy = torch.randn(b * t, 1024, device=dev).to(torch.bfloat16)
        res = []
        for fused in (True, False):
            os.environ["DVQ_NO_FUSED_ATTN"] = "0" if fused else "1"
            try:
                rt._seed_counter[0] = 77
                for p_ in attn.parameters():
                    p_.grad = None
                tape = Tape()
                y = attn.fwd(x, b, t, tape)
                assert ("fused" in tape.s) == fused
                dx = attn.bwd(dy, tape)
                grads = [sg._grad_buf(p_).clone() for p_ in (attn.query.weight, attn.key.weight, attn.value.weight)]
                for p_ in attn.parameters():
                    sg._grad_buf(p_).zero_()
                res.append([y.float().cpu().numpy(), dx.float().cpu().numpy()] + [g.float().cpu().numpy() for g in grads])
            finally:
                os.environ.pop("DVQ_NO_FUSED_ATTN", None)
    for name, a, r in zip(("y", "dx", "dWq", "dWk", "dWv"), *res):
        # two bf16 pipelines of the same arithmetic: L2 agreement (isolated bf16 rounding flips aside)
        err = float(np.linalg.norm(a - r) / max(1e-9, np.linalg.norm(r)))
        assert err < 2e-2, f"{name}: relative L2 difference {err}"


def _reference_full(q, k, v, dout, b, t):
    """fp64 restatement of AttnBlock's attention (modules/diffusionmodules/model.py:173-187): one head of size C, softmax over
    all keys of the image, scale C^-1/2"""
    c = q.shape[1]
    qs, ks, vs = (torch.from_numpy(a).double().view(b, t, c).requires_grad_(True) for a in (q, k, v))
    att = (qs @ ks.transpose(-2, -1)) * (float(c) ** -0.5)
    y = (att.softmax(dim=-1) @ vs).reshape(b * t, c)
    (y * torch.from_numpy(dout).double()).sum().backward()
    return y.detach().numpy(), qs.grad.reshape(b * t, c).numpy(), ks.grad.reshape(b * t, c).numpy(), vs.grad.reshape(b * t, c).numpy()


@pytest.mark.parametrize("shape", [(2, 256), (1, 32), (3, 1024), (2, 160)], ids=lambda s: "x".join(map(str, s)))
def test_fused_full_attention_vs_fp64_reference(dev, shape):
    """single-head full attention with head size 256 (the AttnBlocks at 32x32 / 16x16 maps of 256 channels): forward, dQ, dK, dV"""
    from dynamicvectorquantization_amd import kernels as K
    b, t = shape
    c = 256
    rs = np.random.RandomState(b * 100 + t)
    q, k, v, dout = (_bf16(rs.standard_normal((b * t, c)).astype(np.float32) * s) for s in (2.0, 2.0, 1.0, 1.0))
    yr, dqr, dkr, dvr = _reference_full(q, k, v, dout, b, t)
    dq_, dk_, dv_, do_ = (torch.from_numpy(a).to(dev, torch.bfloat16) for a in (q, k, v, dout))
    assert K.attn_full_ok(dq_, t)
    y, lse = K.attn_full_fwd(dq_, dk_, dv_, b, t, float(c) ** -0.5)
    gq, gk, gv = K.attn_full_bwd(dq_, dk_, dv_, y, do_, lse, b, t, float(c) ** -0.5)
    torch.cuda.synchronize()
    assert _rel(y.float().cpu().numpy(), yr) < 2e-2
    lse_ref = torch.logsumexp((torch.from_numpy(q).double().view(b, t, c) @ torch.from_numpy(k).double().view(b, t, c).transpose(1, 2))
                              * (float(c) ** -0.5), dim=-1).numpy()
    assert float(np.abs(lse.cpu().numpy() - lse_ref).max()) < 2e-2
    assert _rel(gq.float().cpu().numpy(), dqr) < 3e-2 and _rel(gk.float().cpu().numpy(), dkr) < 3e-2
    assert _rel(gv.float().cpu().numpy(), dvr) < 3e-2


def test_attnblock_fused_equals_unfused_path(dev, monkeypatch):
    """AttnBlock (C = 256, 32x32 map): the fused path and the GEMM + softmax path agree on output, input gradient and
    parameter gradients (bf16 tolerance); C = 512 blocks stay on the GEMM path"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd.layers import AttnBlock
    res = {}
    with rt.compute_dtype_ctx(torch.bfloat16):
        for tag, off in (("fused", "0"), ("gemm", "1")):
            monkeypatch.setenv("DVQ_NO_FUSED_ATTNBLOCK", off)
            torch.manual_seed(0)
            blk = AttnBlock(256).to(dev)
            x = torch.randn(2, 256, 32, 32, device=dev).requires_grad_(True)
            y = blk(x)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            res[tag] = (y.detach().float(), x.grad.float(), blk.q.weight.grad.float().clone(), blk.v.weight.grad.float().clone(),
                        blk.proj_out.weight.grad.float().clone())
        assert not K.attn_full_ok(torch.empty(64, 512, dtype=torch.bfloat16, device=dev), 64)
    for a, b_ in zip(res["fused"], res["gemm"]):
        assert float((a - b_).abs().max()) <= 3e-2 * float(b_.abs().max()) + 1e-6


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_fused_qkv_projection_equals_three_linears(dev, monkeypatch, p_drop):
    """CausalSelfAttention (stackgpt.py:41-69 of the reference: three Linear layers over the same input) with key / query / value as ONE
    GEMM over row-concatenated weights, the attention kernels reading column blocks of its [M, 3 C] output with a row pitch and ONE
    input-gradient GEMM over K = 3 C (DVQ_QKV_FUSED=1, the default for head size 128) against the three-Linear path of the same module:
    output, input gradient and the six parameter gradients; then a second step after a parameter change (the fused operand must follow
    the repack)"""
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import stackgpt as sg
    from dynamicvectorquantization_amd.layers import Tape
    torch.manual_seed(5)
    cfg = sg.StackGPTConfig(n_embd=256, n_head=2, block_size=128, attn_pdrop=p_drop, resid_pdrop=0.0)
    attn = sg.CausalSelfAttention(cfg).to(dev)
    attn.train()
    b, t, c = 3, 72, 256
    x = (torch.randn(b * t, c, device=dev) * 0.7).to(torch.bfloat16)
    dy = torch.randn(b * t, c, device=dev).to(torch.bfloat16)
    names = [n for n, _ in attn.named_parameters()]

    def run(fused):
        monkeypatch.setenv("DVQ_QKV_FUSED", "1" if fused else "0")
        for p_ in attn.parameters():
            p_.grad = torch.zeros_like(p_)
        rt._seed_counter[0] = 1000                       # the same dropout seeds in both runs
        with rt.compute_dtype_ctx(torch.bfloat16):
            tape = Tape()
            y = attn.fwd(x, b, t, tape)
            assert ("qkv" in tape.s) == fused
            dx = attn.bwd(dy, tape)
        torch.cuda.synchronize()
        return y.float(), dx.float(), {n: p_.grad.clone() for n, p_ in attn.named_parameters()}

    rel = lambda a, r: float((a - r).norm() / r.norm().clamp_min(1e-20))
    for step in range(2):
        y1, dx1, g1 = run(True)
        y0, dx0, g0 = run(False)
        assert rel(y1, y0) < 1e-2 and rel(dx1, dx0) < 2e-2, (step, rel(y1, y0), rel(dx1, dx0))
        for n in names:
            assert rel(g1[n], g0[n]) < 2e-2, (step, n, rel(g1[n], g0[n]))
        with torch.no_grad():                            # an optimizer step: every packed copy must follow
            for p_ in attn.parameters():
                p_.add_(0.05 * torch.randn_like(p_))
        rt.bump_weights_epoch()
