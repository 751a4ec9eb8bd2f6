#!/usr/bin/env python3
"""conv_halo2.hip (persistent 3x3 kernel) against conv3x3_halo_kernel: same inputs, every epilogue combination the persistent kernel
takes, several shapes (incl. Cout = 256 / 512, nearest-x2 input, the flipped weight pack); then timings of both at 64 x 256^2 x 128.
Child processes: DVQ_HALO2 is read once per process."""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(mode):
    import torch
    from dynamicvectorquantization_amd import kernels as K, runtime as rt
    from dynamicvectorquantization_amd.layers import Conv2d
    dev = torch.device("cuda:0")
    rt.set_compute_dtype(torch.bfloat16)
    K.ensure_workspace(dev)
    out = {}
    shapes = [(8, 64, 64, 128, 128), (4, 32, 64, 256, 256), (2, 32, 32, 512, 512), (8, 64, 64, 64, 128), (64, 256, 256, 128, 128)]
    if mode == "time":
        shapes = shapes[-1:]
    for (B, H, W, Ci, Co) in shapes:
        torch.manual_seed(1)
        conv = Conv2d(Ci, Co, 3, 1, 1).to(dev)
        with torch.no_grad():
            conv.bias.normal_()
        w, wt, bias = conv.packed(torch.bfloat16)
        g = torch.randn(B, H, W, Ci, device=dev)
        x = (g * torch.sigmoid(g)).to(torch.bfloat16)
        r = torch.randn(B, H, W, Co, device=dev).to(torch.bfloat16)
        d = conv._desc(x)
        tag = f"{B}x{H}x{W}x{Ci}->{Co}"
        def stats_call(res):
            st = torch.zeros(B, 32, 2, dtype=torch.float64, device=dev)
            y = K.conv2d_fwd(d, x, w, bias, res, out_stats=st, out_groups=32)
            return y, st
        cases = {
            "fwd": lambda: (K.conv2d_fwd(d, x, w, bias, None), None),
            "fwd_nobias": lambda: (K.conv2d_fwd(d, x, w, None, None), None),
            "fwd_res": lambda: (K.conv2d_fwd(d, x, w, bias, r), None),
            "fwd_stats": lambda: stats_call(None),
            "fwd_res_stats": lambda: stats_call(r),
            "fwd_relu": lambda: (K.conv2d_fwd(d, x, w, bias, None, act=K.ACT_RELU), None),
        }
        if Ci == Co:
            cases["dgrad"] = lambda: (K.conv2d_dgrad(d, r, wt), None)
            cases["dgrad_gate"] = lambda: (K.conv2d_dgrad(d, r, wt, mask=x, mask_act=K.ACT_RELU), None)
        for name, fn in cases.items():
            if mode == "time":
                for _ in range(2): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                ms = s.elapsed_time(e) / 10
                out[f"{tag} {name}"] = [round(ms, 4), round(2 * B * H * W * Ci * Co * 9 / ms / 1e9)]
            else:
                y, st = fn()
                y2, st2 = fn()                                     # run to run: bit-identical
                torch.cuda.synchronize()
                rec = {"sum": float(y.float().double().sum()), "abs": float(y.float().double().abs().sum()), "rr": bool(torch.equal(y, y2))}
                if st is not None:
                    rec["stats"] = [float(st[..., 0].sum()), float(st[..., 1].sum())]
                    rec["stats_row"] = st[B // 2, 5].tolist()
                torch.save(y.cpu(), f"/tmp/h2_{os.environ.get('DVQ_HALO2','1')}_{tag}_{name}.pt")
                out[f"{tag} {name}"] = rec
    print("RESULT " + json.dumps(out), flush=True)


def run(halo2, mode):
    env = dict(os.environ, DVQ_HALO2=str(halo2))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", mode], env=env, capture_output=True, text=True, timeout=900)
    for l in r.stdout.splitlines():
        if l.startswith("RESULT "):
            return json.loads(l[7:])
    raise RuntimeError(r.stdout[-2000:] + r.stderr[-3000:])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(sys.argv[2])
        sys.exit(0)
    import torch
    if os.environ.get("H2_SKIP_CHECK", "0") != "1":
        old, new = run(0, "check"), run(2, "check")
        bad = 0
        for k in old:
            tag, name = k.split(" ")
            a = torch.load(f"/tmp/h2_0_{tag}_{name}.pt").float()
            b = torch.load(f"/tmp/h2_2_{tag}_{name}.pt").float()
            diff = (a - b).abs()
            rel = float(diff.norm() / (a.norm() + 1e-30))
            nbad = int((diff > 0.02 * a.abs() + 0.02).sum())
            st = ""
            if "stats" in old[k]:
                so, sn = old[k]["stats"], new[k]["stats"]
                st = f" stats {so[0]:.4f}/{sn[0]:.4f} {so[1]:.4f}/{sn[1]:.4f} row {old[k]['stats_row']} / {new[k]['stats_row']}"
                if abs(so[0] - sn[0]) > 1e-3 * abs(so[1]) ** 0.5 + 1e-3 * abs(so[0]) or abs(so[1] - sn[1]) > 1e-3 * abs(so[1]):
                    bad += 1
                    st += " STATS-MISMATCH"
            flag = "" if (rel < 3e-3 and nbad == 0 and new[k]["rr"]) else "  <<<<<< MISMATCH"
            bad += flag != ""
            print(f"{k:44s} rel {rel:.2e} outliers {nbad} equal {bool(torch.equal(a, b))} run-to-run {new[k]['rr']}{st}{flag}", flush=True)
        print("CHECK", "FAILED" if bad else "OK", bad)
    told, tnew = run(0, "time"), run(1, "time")
    for k in told:
        print(f"{k:44s} old {told[k][0]:.4f} ms {told[k][1]} TF/s   new {tnew[k][0]:.4f} ms {tnew[k][1]} TF/s   x{told[k][0] / tnew[k][0]:.3f}", flush=True)
    for dbg in os.environ.get("H2_DBG", "").split():
        os.environ["DVQ_HALO2_DBG"] = dbg
        t = run(1, "time")
        k = next(iter(t))
        print(f"DVQ_HALO2_DBG={dbg}: {t[k][0]:.4f} ms {t[k][1]} TF/s", flush=True)
