#!/usr/bin/env python3
"""Attribute the in-step penalty of the halo convolution with COUNTERS: one eager training step of the headline
config runs with every 3x3 / 128 -> 128 / 256^2 forward and input-gradient launch captured (arguments cloned), then the same launches
are replayed back to back, three times over, in the same process.  Run under `rocprofv3 --pmc <counters> --kernel-trace`: the
post-processing (`--summarise <counter_collection.csv> <kernel_trace.csv>`) pairs launch i of the step with its three replays.

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace \
        --output-format csv -d out -o p -- python tools/probes/halo_instep_pmc.py
    python tools/probes/halo_instep_pmc.py --summarise out/.../p_counter_collection.csv out/.../p_kernel_trace.csv
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def summarise(cc_path, kt_path):
    import collections
    import csv
    dur = {}
    for r in csv.DictReader(open(kt_path)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
    ctr = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc_path)):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = [d for d in sorted(dur, key=lambda x: int(x)) if "conv3x3_halo_kernel<4, false, false>" in dur[d][1] and dur[d][2] == 16384 * 256]
    kinds = open(os.path.join(REPO, "gpurun_out", "halo_instep_kinds.txt")).read().split()
    n = len(kinds)
    assert len(ids) >= 4 * n, (len(ids), n)
    step, reps = ids[-4 * n:-3 * n], [ids[-3 * n + k * n: -3 * n + (k + 1) * n] if k < 2 else ids[-n:] for k in range(3)]
    names = sorted({c for d in ids for c in ctr[d]})
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for i, kind in enumerate(kinds):
        cnt[kind] += 1
        agg[kind]["ns_step"] += dur[step[i]][0]
        agg[kind]["ns_replay"] += sum(dur[r[i]][0] for r in reps) / 3.0
        for c in names:
            agg[kind][c + "_step"] += ctr[step[i]].get(c, 0.0)
            agg[kind][c + "_replay"] += sum(ctr[r[i]].get(c, 0.0) for r in reps) / 3.0
    print(f"{'variant':14s}{'launches':>9s}{'ms/launch step':>16s}{'replay':>10s}{'ratio':>8s}   " + "   ".join(f"{c} step/replay" for c in names))
    for kind in sorted(agg):
        a = agg[kind]
        k = cnt[kind]
        row = f"{kind:14s}{k:9d}{a['ns_step'] / k / 1e6:16.4f}{a['ns_replay'] / k / 1e6:10.4f}{a['ns_step'] / a['ns_replay']:8.3f}   "
        row += "   ".join(f"{a[c + '_step'] / max(1.0, a[c + '_replay']):.4f}" for c in names)
        print(row)


if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2], sys.argv[3])
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
from dynamicvectorquantization_amd import _lib, kernels as K, runtime as rt, synth  # noqa: E402
from dynamicvectorquantization_amd.config import instantiate_from_config  # noqa: E402
from dynamicvectorquantization_amd.trainer import Trainer, reference_learning_rate  # noqa: E402

dev = torch.device("cuda", 0)
_lib.check(_lib.load().dvq_check_device(), "dvq_check_device")
rt.set_compute_dtype("bf16")
torch.manual_seed(0)
BS = 64
model = instantiate_from_config(bench.full_config("full", BS)).to(dev)
model.learning_rate = reference_learning_rate({"base_learning_rate": 4.5e-6}, 1, BS)
model.training_steps, model.steps_per_epoch = 100000, 1000
model.train()
os.environ["DVQ_SIDE_WGRAD"] = "0"          # one stream: the launches of the step are then ordered as issued
tr = Trainer(model, max_steps=8, use_graph=False)
batches = [{"image": torch.from_numpy(synth.half_flat_images(BS, 256, seed=1234 + 1000 * i)).to(dev)} for i in range(2)]
for i in range(2):
    tr.train_step(batches[i % 2], i)
torch.cuda.synchronize()
cap = []
MAXCAP = int(os.environ.get("MAXCAP", "14"))
of, od = K.conv2d_fwd, K.conv2d_dgrad


def match(d):
    return (d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.stride, d.upsample) == (BS, 256, 256, 128, 128, 3, 1, 0)


def spy_f(d, x, w, bias, residual=None, gn_ss=None, out_stats=None, out_groups=0, act=K.ACT_NONE):
    if match(d) and act == K.ACT_NONE and len(cap) < MAXCAP:
        kind = "fwd" + ("+res" if residual is not None else "") + ("+gn" if gn_ss is not None else "") + ("+st" if out_stats is not None else "")
        cap.append((kind, dict(d=d, x=x.clone(), w=w.clone(), bias=None if bias is None else bias.clone(), residual=None if residual is None else residual.clone(),
                               gn_ss=None if gn_ss is None else gn_ss.clone(), groups=out_groups, stats=out_stats is not None)))
    return of(d, x, w, bias, residual, gn_ss=gn_ss, out_stats=out_stats, out_groups=out_groups, act=act)


def spy_d(d, dy, wt, *a, **k):
    if match(d) and len(cap) < MAXCAP:
        cap.append(("dgrad" + ("+mask" if (a or k) else ""), dict(d=d, dy=dy.clone(), wt=wt.clone(), a=tuple(t.clone() if torch.is_tensor(t) else t for t in a),
                                                                  k={n: (t.clone() if torch.is_tensor(t) else t) for n, t in k.items()})))
    return od(d, dy, wt, *a, **k)


K.conv2d_fwd, K.conv2d_dgrad = spy_f, spy_d
import dynamicvectorquantization_amd.layers as L  # noqa: E402,F401
tr.train_step(batches[0], 2)
K.conv2d_fwd, K.conv2d_dgrad = of, od
torch.cuda.synchronize()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "halo_instep_kinds.txt"), "w").write(" ".join(k for k, _ in cap))
del tr, model
for _ in range(3):
    for kind, c in cap:
        if kind.startswith("dgrad"):
            od(c["d"], c["dy"], c["wt"], *c["a"], **c["k"])
        else:
            st = K.zeros_small((c["d"].N, 32, 2), torch.float64, dev) if c["stats"] else None
            of(c["d"], c["x"], c["w"], c["bias"], c["residual"], gn_ss=c["gn_ss"], out_stats=st, out_groups=c["groups"] if c["stats"] else 0)
    torch.cuda.synchronize()
print("captured", len(cap), [k for k, _ in cap])
