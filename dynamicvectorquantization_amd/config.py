"""The drop-in boundary: the reference's `target:` / `params:` plugin resolver and YAML + dotlist config
loading (utils/utils.py:41-51, train.py:109-111), without OmegaConf.

`instantiate_from_config({"target": "pkg.mod.Class", "params": {...}})` resolves the reference's own
dotted paths (modules.vector_quantization.quantize2_mask.VectorQuantize2, ...) to this package's classes
through TARGET_ALIASES, so the shipped YAMLs work unchanged; any other target is imported normally.
`install_reference_aliases()` additionally registers synthetic modules under the reference's names in
sys.modules so `from modules.dynamic_modules.EncoderDual import DualGrainEncoder` keeps working.
"""
from __future__ import annotations

import copy
import importlib
import sys
import types

import yaml

_P = "dynamicvectorquantization_amd."

# reference dotted path -> (module in this package, attribute)
TARGET_ALIASES = {
    "models.stage1_dynamic.dqvae_dual_entropy.DualGrainVQModel": (_P + "dqvae", "DualGrainVQModel"),
    "models.stage1_dynamic.dqvae_dual_entropy.Entropy": (_P + "dqvae", "Entropy"),
    "models.stage1_dynamic.dqvae_dual_feat.DualGrainVQModel": (_P + "dqvae", "DualGrainFeatVQModel"),
    "models.stage1_dynamic.dqvae_triple_feat.TripleGrainVQModel": (_P + "dqvae", "TripleGrainVQModel"),
    "modules.dynamic_modules.EncoderTriple.TripleGrainEncoder": (_P + "dqvae", "TripleGrainEncoder"),
    "modules.dynamic_modules.RouterTriple.TripleGrainFeatureRouter": (_P + "dqvae", "TripleGrainFeatureRouter"),
    "modules.dynamic_modules.EncoderDual.DualGrainEncoder": (_P + "dqvae", "DualGrainEncoder"),
    "modules.dynamic_modules.RouterDual.DualGrainFixedEntropyRouter": (_P + "dqvae", "DualGrainFixedEntropyRouter"),
    "modules.dynamic_modules.RouterDual.DualGrainFeatureRouter": (_P + "dqvae", "DualGrainFeatureRouter"),
    "modules.dynamic_modules.DecoderPositional.Decoder": (_P + "dqvae", "Decoder"),
    "modules.dynamic_modules.DecoderPositional.PositionEmbedding2DLearned": (_P + "dqvae", "PositionEmbedding2DLearned"),
    "modules.dynamic_modules.fourier_embedding.FourierPositionEmbedding": (_P + "dqvae", "FourierPositionEmbedding"),
    "modules.vector_quantization.quantize2_mask.VectorQuantize2": (_P + "quantize", "VectorQuantize2"),
    "modules.vector_quantization.quantize2_mask.VQEmbedding": (_P + "quantize", "VQEmbedding"),
    "modules.diffusionmodules.model.ResnetBlock": (_P + "layers", "ResnetBlock"),
    "modules.diffusionmodules.model.AttnBlock": (_P + "layers", "AttnBlock"),
    "modules.diffusionmodules.model.Upsample": (_P + "layers", "Upsample"),
    "modules.diffusionmodules.model.Downsample": (_P + "layers", "Downsample"),
    "modules.diffusionmodules.model.Normalize": (_P + "layers", "Normalize"),
    "modules.diffusionmodules.model.nonlinearity": (_P + "layers", "nonlinearity"),
    "modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator": (_P + "losses", "VQLPIPSWithDiscriminator"),
    "modules.losses.vqperceptual.DummyLoss": (_P + "losses", "DummyLoss"),
    "modules.discriminator.model.NLayerDiscriminator": (_P + "losses", "NLayerDiscriminator"),
    "modules.dynamic_modules.budget.BudgetConstraint_RatioMSE_DualGrain": (_P + "losses", "BudgetConstraint_RatioMSE_DualGrain"),
    "modules.dynamic_modules.budget.BudgetConstraint_NormedSeperateRatioMSE_TripleGrain":
        (_P + "losses", "BudgetConstraint_NormedSeperateRatioMSE_TripleGrain"),
    "models.stage1.utils.Scheduler_LinearWarmup": (_P + "trainer", "scheduler_linear_warmup"),
    "models.stage1.utils.Scheduler_LinearWarmup_CosineDecay": (_P + "trainer", "scheduler_linear_warmup_cosine_decay"),
    "modules.dynamic_modules.stackgpt.StackGPT": (_P + "stackgpt", "StackGPT"),
    "models.stage2_dynamic.dqtransformer_uncond_entropy.Dualformer": (_P + "stage2", "Dualformer"),
    "models.stage2_dynamic.dqtransformer_class2_entropy.Dualformer": (_P + "stage2", "ClassDualformer"),
    "models.stage2.utils.learning_rate_schedule": (_P + "trainer", "scheduler_linear_warmup_cosine_decay"),
    "modules.dynamic_modules.permuter.DualGrainSeperatePermuter": (_P + "stage2", "DualGrainSeperatePermuter"),
    "modules.dynamic_modules.label_provider.PositionAwareSOSProvider": (_P + "stage2", "PositionAwareSOSProvider"),
    "modules.dynamic_modules.label_provider.ClassForContentOnlyPositionAwareSOSProvider":
        (_P + "stage2", "ClassForContentOnlyPositionAwareSOSProvider"),
    "modules.dynamic_modules.label_provider.ClassAwareSOSProvider": (_P + "stage2", "ClassAwareSOSProvider"),
    "utils.utils.instantiate_from_config": (_P + "config", "instantiate_from_config"),
    "data.build.DataModuleFromConfig": (_P + "data", "DataModuleFromConfig"),
    "data.imagenet.ImageNetTrain": (_P + "data", "ImageNetTrain"),
    "data.imagenet.ImageNetValidation": (_P + "data", "ImageNetValidation"),
    "data.imagenet_base.ImagePaths": (_P + "data", "ImagePaths"),
}


def get_obj_from_str(string, reload=False):
    if string in TARGET_ALIASES:
        mod, attr = TARGET_ALIASES[string]
        return getattr(importlib.import_module(mod), attr)
    module, cls = string.rsplit(".", 1)
    module_imp = importlib.import_module(module)
    if reload:
        importlib.reload(module_imp)
    return getattr(module_imp, cls)


def instantiate_from_config(config):
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params", dict()) or dict()))


def install_reference_aliases():
    """Register synthetic modules under the reference's dotted names (idempotent)."""
    for dotted, (mod, attr) in TARGET_ALIASES.items():
        modname, name = dotted.rsplit(".", 1)
        parts = modname.split(".")
        for i in range(1, len(parts) + 1):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []          # behave like a package
                m.__dvq_alias__ = True
                sys.modules[pkg] = m
                if i > 1:
                    setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
        target = sys.modules[modname]
        if getattr(target, "__dvq_alias__", False):
            try:
                setattr(target, name, getattr(importlib.import_module(mod), attr))
            except Exception:      # optional pieces (e.g. trainer extras) must not break the others
                pass


# ---------------------------------------------------------------------------------------------
# YAML + dotlist (train.py:109-111 uses OmegaConf.load / from_dotlist / merge)
# ---------------------------------------------------------------------------------------------
class Cfg(dict):
    """dict with attribute access (enough of OmegaConf's DictConfig for train.py)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def load_yaml(path) -> Cfg:
    with open(path, "r", encoding="utf-8") as f:
        return _wrap(yaml.safe_load(f) or {})


def _parse_scalar(s: str):
    try:
        return yaml.safe_load(s)
    except yaml.YAMLError:
        return s


def from_dotlist(items) -> Cfg:
    out = Cfg()
    for it in items:
        if "=" not in it:
            raise ValueError(f"override `{it}` is not of the form key.sub=value")
        key, val = it.split("=", 1)
        cur = out
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, Cfg())
        cur[parts[-1]] = _wrap(_parse_scalar(val))
    return out


def merge(*cfgs) -> Cfg:
    def _m(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                _m(a[k], v)
            else:
                a[k] = copy.deepcopy(v)
        return a

    out = Cfg()
    for c in cfgs:
        _m(out, _wrap(c))
    return out


def to_plain(o):
    if isinstance(o, dict):
        return {k: to_plain(v) for k, v in o.items()}
    if isinstance(o, list):
        return [to_plain(v) for v in o]
    return o


# ---------------------------------------------------------------------------------------------
# the shipped stage-1 YAML as the single source of a DQ-VAE's constructor arguments (bench.py, __graft_entry__.smoke)
# ---------------------------------------------------------------------------------------------
REPO_ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
STAGE1_DUAL_ENTROPY_YAML = "configs/stage1/dqvae-entropy-dual-r05_imagenet.yml"


def stage1_config(yaml_path=STAGE1_DUAL_ENTROPY_YAML, dotlist=(), batch_size=None, objective="full", geometry=None) -> Cfg:
    """`train.py --base <yaml> key=value ...` as a function (train.py:109-111): load the YAML, merge the dotlist.

    batch_size : the one override a benchmark applies (`data.params.batch_size`)
    objective  : "full" = the file's two-optimizer objective; "ae" = its LPIPS and GAN terms switched off
                 (perceptual_weight = disc_factor = 0: L1 + codebook); "none" = modules.losses.vqperceptual.DummyLoss
    geometry   : dict(ch, resolution, latent, zc, k, attn_enc, attn_dec) -- the shrunken / 64 x 64 variants the golden fixtures and
                 smoke() use; every other value stays the file's
    """
    import os
    path = yaml_path if os.path.isabs(yaml_path) else os.path.join(REPO_ROOT, yaml_path)
    dot = list(dotlist)
    if batch_size is not None:
        dot.append(f"data.params.batch_size={int(batch_size)}")
    if objective == "ae":
        dot += ["model.params.lossconfig.params.perceptual_weight=0.0", "model.params.lossconfig.params.disc_factor=0.0"]
    c = merge(load_yaml(path), from_dotlist(dot))
    if objective == "none":
        c.model.params.lossconfig = Cfg(target="modules.losses.vqperceptual.DummyLoss")
    if geometry:
        g, p = geometry, c.model.params
        p.encoderconfig.params.update(ch=g["ch"], resolution=g["resolution"], z_channels=g["zc"], attn_resolutions=list(g["attn_enc"]))
        p.decoderconfig.params.update(ch=g["ch"], in_ch=g["zc"], resolution=g["resolution"], attn_resolutions=list(g["attn_dec"]),
                                      latent_size=g["latent"])
        p.vqconfig.params.update(codebook_size=g["k"], codebook_dim=g["zc"])
        p.quant_before_dim = p.quant_after_dim = g["zc"]
        p.image_size = g["resolution"]
    return c
