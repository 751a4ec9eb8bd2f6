#!/usr/bin/env python3
"""`train.py` with the reference's command-line surface (train.py:27-50,58-270 of the reference) on the HIP
path, without pytorch_lightning / OmegaConf:

    python train.py --gpus -1 --base configs/stage1/dqvae-entropy-dual-r05_imagenet.yml --max_epochs 50 \
        model.params.lossconfig.params.perceptual_weight=0 model.params.lossconfig.params.disc_factor=0

* `-b/--base` YAMLs are merged left to right, then `key.sub=value` overrides (train.py:109-111);
* `--gpus N|-1|a,b,c`: one process per GPU; when launched under torchrun the env ranks are used, otherwise
  ranks are spawned here; gradients are averaged with RCCL (`trainer.GradBuckets`);
* learning rate = ngpu * batch_size * base_learning_rate (train.py:248-257);
* data: the BASELINE configs run on synthetic batches (`--synthetic`, default); a real
  `data:` section is instantiated only when its target is importable.
Unsupported Trainer flags are accepted and ignored with a warning, like unknown Lightning flags would be.
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("-n", "--name", type=str, const=True, default="", nargs="?", help="postfix for logdir")
    p.add_argument("-r", "--resume", type=str, const=True, default="", nargs="?", help="resume from logdir or checkpoint")
    p.add_argument("-b", "--base", nargs="*", metavar="base_config.yaml", default=list())
    p.add_argument("-s", "--seed", type=int, default=2021)
    p.add_argument("-f", "--postfix", type=str, default="")
    p.add_argument("-l", "--logger", type=str, default="none")
    p.add_argument("-d", "--debug", action="store_true")
    p.add_argument("-p", "--project", type=str, default="dvq")
    p.add_argument("--save_n", type=int, default=3, help="save top-n checkpoints by the model's `monitor` (train.py:48 of the reference)")
    p.add_argument("--check_val_every_n_epoch", type=int, default=1, help="validation pass every n epochs (0: never)")
    p.add_argument("--val_batches", type=int, default=4, help="synthetic data: batches per validation pass")
    p.add_argument("--activate_ddp_share", action="store_true")
    p.add_argument("--gpus", type=str, default="1")
    p.add_argument("--max_epochs", type=int, default=1)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--steps_per_epoch", type=int, default=100, help="synthetic data: batches per epoch")
    p.add_argument("--synthetic", action="store_true", default=True)
    p.add_argument("--real_data", action="store_true",
                   help="instantiate the config's `data:` section (data.build.DataModuleFromConfig -> GPU input pipeline, "
                        "$DVQ_IMAGENET_ROOT/train|val) instead of synthetic batches")
    p.add_argument("--precision", type=str, default="bf16", help="compute dtype of the HIP path: bf16 | fp32 | fp32x3 (fp32 tensors, products as three bf16 MFMA passes on split operands)")
    p.add_argument("--logdir", type=str, default="logs")
    p.add_argument("--save_every", type=int, default=0,
                   help="rewrite checkpoints/last.ckpt every N steps (0: once per epoch) -- atomic, rank 0 only")
    return p


def _gpu_ids(spec: str, n_visible: int):
    """--gpus: "-1" = all visible devices, "N" = devices 0..N-1, "a,b,c" = exactly those device ids (train.py:61-62, Lightning)"""
    spec = spec.strip()
    if spec == "-1":
        return list(range(n_visible))
    if "," in spec:
        return [int(g) for g in spec.split(",") if g.strip() != ""]
    return list(range(int(spec)))


def run(rank, world, opt, unknown):
    import torch
    import torch.distributed as dist
    from dynamicvectorquantization_amd import config as cfg
    from dynamicvectorquantization_amd import runtime as rt
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.trainer import Trainer

    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if "LOCAL_RANK" in os.environ:                    # torchrun: the launcher's local rank is the device
        device_id = int(os.environ["LOCAL_RANK"])
    else:                                             # own spawn: rank r drives the r-th id of --gpus
        ids = _gpu_ids(opt.gpus, torch.cuda.device_count())
        device_id = ids[rank] if rank < len(ids) else rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_id)
    dev = torch.device("cuda", torch.cuda.current_device())
    rt.set_compute_dtype(opt.precision)
    torch.manual_seed(opt.seed)

    dot = [u for u in unknown if "=" in u and not u.startswith("--")]
    ignored = [u for u in unknown if u not in dot]
    if ignored and rank == 0:
        print(f"[train.py] ignoring unsupported Trainer flags: {ignored}")
    # -r <logdir | checkpoint>: continue IN that logdir with the configs it saved (train.py:73-89 of the reference), -b adds to them
    resume_ckpt, logdir = None, None
    if opt.resume:
        if os.path.isfile(opt.resume):
            resume_ckpt = opt.resume
            logdir = os.path.dirname(os.path.dirname(os.path.abspath(opt.resume)))
        else:
            logdir = opt.resume.rstrip("/")
            resume_ckpt = os.path.join(logdir, "checkpoints", "last.ckpt")
        saved_cfgs = sorted(os.path.join(logdir, "configs", f) for f in os.listdir(os.path.join(logdir, "configs"))) \
            if os.path.isdir(os.path.join(logdir, "configs")) else []
        opt.base = saved_cfgs + list(opt.base)
    if not opt.base:
        raise SystemExit("-b/--base <config.yaml> is required (or -r <logdir> holding configs/)")
    config = cfg.merge(*[cfg.load_yaml(b) for b in opt.base], cfg.from_dotlist(dot))
    if logdir is None:
        now = datetime.datetime.now().strftime("%Y-%m-%dT%H-%M-%S")
        logdir = os.path.join(opt.logdir, now + ("_" + opt.name if opt.name else "") + opt.postfix)
    if world > 1:            # every rank must agree on the directory name (timestamps differ): rank 0's wins
        names = [logdir]
        dist.broadcast_object_list(names, src=0)
        logdir = names[0]
    ckpt_path = os.path.join(logdir, "checkpoints", "last.ckpt")
    if rank == 0 and not opt.resume:
        os.makedirs(os.path.join(logdir, "configs"), exist_ok=True)
        import yaml
        with open(os.path.join(logdir, "configs", "project.yaml"), "w") as f:
            yaml.safe_dump(cfg.to_plain(config), f)
    model = cfg.instantiate_from_config(config.model).to(dev)

    bs = config.data.params.batch_size
    model.steps_per_epoch = opt.steps_per_epoch
    model.training_steps = opt.steps_per_epoch * opt.max_epochs
    model.max_epoch = opt.max_epochs
    from dynamicvectorquantization_amd.trainer import reference_learning_rate
    model.learning_rate = reference_learning_rate(config.model, world, bs)
    if "base_learning_rate" in config.model and rank == 0:
        print("Setting learning rate to {:.2e} = {} (num_gpus) * {} (batchsize) * {:.2e} (base_lr)".format(
            model.learning_rate, world, bs, config.model.base_learning_rate))
    model.min_learning_rate = config.model.get("min_learning_rate", 0.)

    size = config.model.params.get("image_size", 256)
    real_iter = None
    if opt.real_data:
        # the YAML's own `data:` section through the plugin boundary: host threads decode, resize / crop / flip / normalise run
        # on the GPU (dynamicvectorquantization_amd/data.py); each rank shuffles with its own seed (weak scaling, bs per GPU).
        # Built BEFORE the Trainer: configure_optimizers() bakes steps_per_epoch / training_steps into the LR schedules
        config.data.params["device"] = str(dev)
        config.data.params["size"] = size
        dm = cfg.instantiate_from_config(config.data)
        loader = dm.train_dataloader()
        loader.rng = __import__("numpy").random.default_rng(opt.seed + 977 * rank)
        opt.steps_per_epoch = model.steps_per_epoch = len(loader)
        model.training_steps = len(loader) * opt.max_epochs

        def _batches():
            while True:
                for b in loader:
                    yield {k: v for k, v in b.items() if torch.is_tensor(v)}      # strings (paths, synsets) stay on the host side
        real_iter = _batches()

    total = model.training_steps if opt.max_steps < 0 else min(opt.max_steps, model.training_steps)
    trainer = Trainer(model, max_steps=total, log_every=10 if rank == 0 else 0)
    pool = [torch.from_numpy(synth.half_flat_images(bs, size, seed=opt.seed + 977 * rank + i)).to(dev) for i in range(4)]

    image_key = getattr(model, "image_key", None) or getattr(model, "first_stage_key", "image")
    n_classes = None
    if getattr(model, "cond_stage_key", None) == "class_label":         # class-conditional stage 2: synthetic labels
        n_classes = int(getattr(model.cond_stage_model, "n_classes", 1000))

    def batch_fn(step):
        model.current_epoch = step // opt.steps_per_epoch
        if real_iter is not None:
            b = next(real_iter)
            return {image_key: b["image"], **({"class_label": b["class_label"]} if "class_label" in b and n_classes is not None else {})}
        batch = {image_key: pool[step % len(pool)]}
        if n_classes is not None:
            g = torch.Generator().manual_seed(opt.seed + step)
            batch["class_label"] = torch.randint(0, n_classes, (bs,), generator=g).to(dev)
        return batch

    if resume_ckpt:
        trainer.load_state_dict(torch.load(resume_ckpt, map_location="cpu", weights_only=False))
        if rank == 0:
            print(f"resumed from {resume_ckpt} at global step {model.global_step}")
    # last.ckpt is rewritten (atomically, rank 0) every --save_every steps / once per epoch and at the end: a crashed or
    # preempted run continues with `-r <logdir>`
    # validation (Lightning's loop in the reference): the YAML's validation set with --real_data, else a fixed set of synthetic batches;
    # the model's `monitor` (val_rec_loss in the shipped stage-1 YAMLs) picks the --save_n best checkpoints kept beside last.ckpt
    val_fn = None
    if opt.check_val_every_n_epoch > 0 and hasattr(model, "validation_step"):
        if opt.real_data and "validation" not in getattr(dm, "datasets", {"validation": None}):
            if rank == 0:
                print("the data config has no validation split: training without the validation loop")
        elif opt.real_data:
            vloader = dm.val_dataloader()

            def val_fn():
                for b in vloader:
                    yield {image_key: b["image"], **({"class_label": b["class_label"]} if "class_label" in b and n_classes is not None else {})}
        else:
            vpool = [torch.from_numpy(synth.half_flat_images(bs, size, seed=opt.seed + 7919 + 977 * rank + i)).to(dev)
                     for i in range(max(1, opt.val_batches))]

            def val_fn():
                for i, im in enumerate(vpool):
                    b = {image_key: im}
                    if n_classes is not None:
                        g = torch.Generator().manual_seed(opt.seed + 31 * i)
                        b["class_label"] = torch.randint(0, n_classes, (bs,), generator=g).to(dev)
                    yield b
    trainer.fit(batch_fn, ckpt_path=ckpt_path, save_every=opt.save_every or opt.steps_per_epoch, is_rank0=rank == 0, val_fn=val_fn,
                val_every=opt.check_val_every_n_epoch * opt.steps_per_epoch, save_top_k=opt.save_n)
    if rank == 0:
        print("saved", ckpt_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    opt, unknown = get_parser().parse_known_args()
    if not opt.base and not opt.resume:
        raise SystemExit("-b/--base <config.yaml> is required")
    import torch
    if "WORLD_SIZE" in os.environ:          # launched by torchrun
        return run(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), opt, unknown)
    ngpu = len(_gpu_ids(opt.gpus, torch.cuda.device_count()))
    if ngpu <= 1:
        return run(0, 1, opt, unknown)
    import torch.multiprocessing as mp
    mp.spawn(run, args=(ngpu, opt, unknown), nprocs=ngpu, join=True)


if __name__ == "__main__":
    main()
