import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden
from dynamicvectorquantization_amd import synth, kernels as K, runtime as rt
from dynamicvectorquantization_amd.config import instantiate_from_config
from test_gpu_lossnet import _toy_last_layer, _load_det, _rel
from oracle import losses as olo
dev = torch.device("cuda:0")
g = load_golden("lossnet")
x = torch.from_numpy(synth.half_flat_images(2, 64, 16, seed=5)).to(dev)
qloss = torch.tensor(0.123, device=dev)
dtype = torch.float32 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
with rt.compute_dtype_ctx(dtype):
    loss_mod = instantiate_from_config({"target": "modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator", "params": dict(
        disc_start=0, disc_init=True, disc_conditional=False, disc_loss="hinge", disc_factor=1.0, disc_weight=1.0,
        disc_weight_max=None, codebook_weight=1.0, pixelloss_weight=1.0, perceptual_weight=1.0,
        disc_config={"target": "modules.discriminator.model.NLayerDiscriminator",
                     "params": dict(input_nc=3, ndf=16, n_layers=3, use_actnorm=False)})}).to(dev).train()
    _load_det(loss_mod.discriminator, "disc.")
    _load_det(loss_mod.perceptual_loss, "lpips.", synth.det_lpips_param)
    conv, tape, xrec = _toy_last_layer(dev, dtype, g)
    # oracle pieces on CPU with the same xrec
    sd_l = {k: v.detach().cpu().float() for k, v in loss_mod.perceptual_loss.state_dict().items()}
    sd_d = {k: v.detach().cpu().float() for k, v in loss_mod.discriminator.state_dict().items()}
    xr = xrec.detach().cpu().requires_grad_(True)
    xc = x.cpu()
    p = olo.lpips(sd_l, xc, xr); gp = torch.autograd.grad(p.mean(), xr)[0]
    l1 = (xc - xr).abs().mean(); gl = torch.autograd.grad(l1, xr)[0]
    gg = torch.autograd.grad(-olo.patchgan(sd_d, xr).mean(), xr)[0]
    out = loss_mod._generator(x, xrec.detach(), True, conv.weight, 1.0)
    cp = K.vec(dtype) * -(-3 // K.vec(dtype))
    x_p = K.nchw_to_nhwc_pad(x, cp, dtype); r_p = K.nchw_to_nhwc_pad(xrec.detach(), cp, dtype)
    val, d_r = loss_mod.perceptual_loss.fwd(x_p, r_p, gscale=0.5)
    d_r = K.nhwc_pad_to_nchw(d_r, 3).cpu()
    print("lpips val", val.cpu().numpy(), p.detach().reshape(-1).numpy())
    print("lpips grad max/l2", _rel(d_r.numpy(), gp.numpy()), _rel(d_r.numpy(), gp.numpy(), True))
    from dynamicvectorquantization_amd.layers import Tape
    t = Tape(); lf = loss_mod.discriminator.fwd(r_p, t)
    dl = torch.zeros_like(lf); dl[..., 0] = -1.0 / lf[..., 0].numel()
    g_g = K.nhwc_pad_to_nchw(loss_mod.discriminator.bwd(dl, t, need_dw=False), 3).cpu()
    print("g_g max/l2", _rel(g_g.numpy(), gg.numpy()), _rel(g_g.numpy(), gg.numpy(), True))
    tot = gl + gp + float(out["d_weight"]) * gg
    print("d_weight", float(out["d_weight"]), float(g["gen_free_d_weight"]))
    print("g_rec max/l2", _rel(out["g_rec"].cpu().numpy(), tot.numpy()), _rel(out["g_rec"].cpu().numpy(), tot.numpy(), True))
    print("norms: gl %.3e gp %.3e dw*gg %.3e" % (gl.norm(), gp.norm(), float(out["d_weight"]) * gg.norm()))
