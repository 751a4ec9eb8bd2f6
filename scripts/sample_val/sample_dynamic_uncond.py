#!/usr/bin/env python3
"""Unconditional sampling with a DQ-Transformer -- the reference's scripts/sample_val/sample_dynamic_uncond.py (:21-118) on the
HIP path (K/V-cached sampler, captured token-step graphs): same flags, same output layout
(`<model>_<time>_Num-<n>/[fixed_]TopK-<k>-<kp>_TopP-<p>-<pp>_Temp-<t>_{pickle,image}/samples_(<i>_<total>).pkl`, pickled
numpy [B,3,H,W] in [0,1]).  --model_path may be omitted (random weights: plumbing / throughput runs); --out_dir overrides
the directory derived from the checkpoint name.

    python scripts/sample_val/sample_dynamic_uncond.py --yaml_path configs/stage2/uncond_imagenet_p6c18.yml \\
        --model_path last.ckpt --batch_size 50 --sample_num 5000 --top_k 300 --top_k_pos 1024 --save_image
"""
import argparse
import datetime
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per sampling lane (bench_extra.py has the measurements); sampling only


def save_pickle(fname, data):
    with open(fname, "wb") as fp:
        pickle.dump(data, fp, pickle.HIGHEST_PROTOCOL)


def save_image_grid(x, path, nrow=8, padding=2):
    """torchvision.utils.save_image for [N,3,H,W] in [0,1] (zero padding between tiles)"""
    import numpy as np
    from PIL import Image
    n, c, h, w = x.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = np.zeros((c, rows * (h + padding) + padding, cols * (w + padding) + padding), dtype=np.float32)
    for i in range(n):
        r, q = divmod(i, cols)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = x[i]
    Image.fromarray((grid.transpose(1, 2, 0) * 255.0 + 0.5).clip(0, 255).astype("uint8")).save(path)


def get_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--yaml_path", type=str, default="")
    parser.add_argument("--model_path", type=str, default="")
    parser.add_argument("--sample_with_fixed_pos", action="store_true", default=False)
    parser.add_argument("--save_image", action="store_true", default=False)
    parser.add_argument("--batch_size", type=int, default=50)
    parser.add_argument("--temperature", type=float, default=1.0)
    parser.add_argument("--top_k", type=int, default=300)
    parser.add_argument("--top_k_pos", type=int, default=1024)
    parser.add_argument("--top_p", type=float, default=1.0)
    parser.add_argument("--top_p_pos", type=float, default=1.0)
    parser.add_argument("--sample_num", type=int, default=5000)
    parser.add_argument("--out_dir", type=str, default="")
    parser.add_argument("--dtype", type=str, default="bf16")
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument("--streams", type=int, default=4,
                        help="batches sampled concurrently, each on its own HIP stream (Dualformer.sample_many); 1 = one batch at a time")
    return parser


def save_image_normalized(x, path):
    """torchvision.utils.save_image(x, path, normalize=True) for ONE [3,H,W] image: min-max scaled to [0,1]"""
    import numpy as np
    from PIL import Image
    lo, hi = float(x.min()), float(x.max())
    y = (np.clip(x, lo, hi) - lo) / max(hi - lo, 1e-5)
    Image.fromarray((y.transpose(1, 2, 0) * 255.0 + 0.5).clip(0, 255).astype("uint8")).save(path)


def main(per_image_png=False):
    """per_image_png: the layout of the reference's scripts/sample_images/sample_dynamic_uncond.py (:40-102) -- one min-max normalised
    PNG per sample, `batch_<i>_<j>.png`, in the `_image` directory, no pickles"""
    opt, _ = get_parser().parse_known_args()
    import time

    import torch
    from dynamicvectorquantization_amd import config as cfg, runtime as rt
    now = datetime.datetime.utcnow().strftime("%m-%dT%H-%M-%S")
    base = opt.out_dir or (opt.model_path.replace(".ckpt", "") if opt.model_path else "samples") + "_{}_Num-{}/".format(now, opt.sample_num)
    tag = "TopK-{}-{}_TopP-{}-{}_Temp-{}".format(opt.top_k, opt.top_k_pos, opt.top_p, opt.top_p_pos, opt.temperature)
    if opt.sample_with_fixed_pos:
        tag = "fixed_" + tag
    dir_img, dir_pkl = os.path.join(base, tag + "_image"), os.path.join(base, tag + "_pickle")
    if opt.save_image or per_image_png:
        os.makedirs(dir_img, exist_ok=True)
    if not per_image_png:
        os.makedirs(dir_pkl, exist_ok=True)

    rt.set_compute_dtype(opt.dtype)
    if opt.seed is not None:
        torch.manual_seed(opt.seed)
    model = cfg.instantiate_from_config(cfg.load_yaml(opt.yaml_path).model)
    if opt.model_path:
        sd = torch.load(opt.model_path, map_location="cpu")
        model.load_state_dict(sd["state_dict"] if "state_dict" in sd else sd)
    model = model.eval().cuda()

    total_batch = (opt.sample_num + opt.batch_size - 1) // opt.batch_size
    steps, t0 = 0, time.perf_counter()
    sizes = [opt.batch_size] * total_batch
    if opt.sample_num % opt.batch_size != 0:
        sizes[-1] = opt.sample_num % opt.batch_size
    kw = dict(temperature=opt.temperature, sample=True, top_k=opt.top_k, top_p=opt.top_p, top_k_pos=opt.top_k_pos, top_p_pos=opt.top_p_pos,
              process=False, fix_fine_position=opt.sample_with_fixed_pos)
    group = max(1, opt.streams) * 4                 # batches handed to the sampler at a time (their token sequences are small)
    with torch.no_grad():
        for g0 in range(0, total_batch, group):
            idxs = [i for i in range(g0, min(total_batch, g0 + group))]
            # full batches go through the concurrent lanes together; a ragged last batch (another cache geometry) runs on its own
            full = [i for i in idxs if sizes[i] == opt.batch_size]
            conds = {i: model.encode_to_c(torch.randn(sizes[i], device="cuda")) for i in idxs}
            outs = dict(zip(full, model.sample_many([conds[i] for i in full], n_streams=opt.streams, **kw))) if full else {}
            for i in idxs:
                if i not in outs:
                    outs[i] = model.sample_from_scratch(*conds[i], **kw)
            for i in idxs:
                batch_size, seqs = sizes[i], outs[i]
                steps += batch_size * int(seqs[0].shape[1] + seqs[1].shape[1])
                raw = model.decode_to_img(*seqs).float()
                if per_image_png:
                    raw = raw.cpu().numpy()
                    for j in range(batch_size):
                        save_image_normalized(raw[j], os.path.join(dir_img, "batch_{}_{}.png".format(i, j)))
                    continue
                img = torch.clamp(raw * 0.5 + 0.5, 0, 1).cpu().numpy()
                if opt.save_image:
                    save_image_grid(img, os.path.join(dir_img, "batch_{}.png".format(i)))
                save_pickle(os.path.join(dir_pkl, "samples_({}_{}).pkl".format(i, total_batch)), img)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("sampled {} images, {} token steps in {:.2f} s ({:.0f} token-steps/s) -> {}".format(opt.sample_num, steps, dt, steps / dt, dir_img if per_image_png else dir_pkl))


if __name__ == "__main__":
    main()
