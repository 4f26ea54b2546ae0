"""The reference's plugin mechanism, served from this package.

The reference builds everything from YAML `target:` strings via `instantiate_from_config` (sgm/util.py:168-185).
`TARGET_MAP` maps every dotted path options/SUPIR_v0*.yaml names on the hot path to the class in this package that
stands in for it; `instantiate_from_config` resolves through it, and `install()` registers alias modules under the
reference's module names so that an unmodified `test.py` / `SUPIR.util.create_SUPIR_model` (which hard-code those
paths) construct the HIP-backed classes.
"""
import importlib
import sys
import types

TARGET_MAP = {
    "SUPIR.models.SUPIR_model.SUPIRModel": "supir_amd.models.supir_model.SUPIRModel",
    "SUPIR.modules.SUPIR_v0.GLVControl": "supir_amd.modules.supir_v0.GLVControl",
    "SUPIR.modules.SUPIR_v0.LightGLVUNet": "supir_amd.modules.supir_v0.LightGLVUNet",
    "SUPIR.modules.SUPIR_v0.ZeroSFT": "supir_amd.modules.supir_v0.ZeroSFT",
    "SUPIR.modules.SUPIR_v0.ZeroCrossAttn": "supir_amd.modules.supir_v0.ZeroCrossAttn",
    "sgm.modules.diffusionmodules.wrappers.ControlWrapper": "supir_amd.modules.wrappers.ControlWrapper",
    "sgm.modules.diffusionmodules.denoiser.DiscreteDenoiserWithControl": "supir_amd.modules.sampling.DiscreteDenoiserWithControl",
    "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting": "supir_amd.modules.sampling.EpsWeighting",
    "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling": "supir_amd.modules.sampling.EpsScaling",
    "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization": "supir_amd.modules.sampling.LegacyDDPMDiscretization",
    "sgm.modules.diffusionmodules.guiders.LinearCFG": "supir_amd.modules.sampling.LinearCFG",
    "sgm.modules.diffusionmodules.guiders.VanillaCFG": "supir_amd.modules.sampling.VanillaCFG",
    "sgm.modules.diffusionmodules.guiders.IdentityGuider": "supir_amd.modules.sampling.IdentityGuider",
    "sgm.modules.diffusionmodules.sampling_utils.NoDynamicThresholding": "supir_amd.modules.sampling.NoDynamicThresholding",
    "sgm.modules.diffusionmodules.sampling.RestoreEDMSampler": "supir_amd.modules.sampling.RestoreEDMSampler",
    "sgm.modules.diffusionmodules.sampling.TiledRestoreEDMSampler": "supir_amd.modules.sampling.TiledRestoreEDMSampler",
    "sgm.modules.diffusionmodules.sampling.RestoreDPMPP2MSampler": "supir_amd.modules.sampling.RestoreDPMPP2MSampler",
    "sgm.modules.diffusionmodules.sampling.TiledRestoreDPMPP2MSampler": "supir_amd.modules.sampling.TiledRestoreDPMPP2MSampler",
    "sgm.modules.diffusionmodules.openaimodel.UNetModel": "supir_amd.modules.openaimodel.UNetModel",
    "sgm.modules.diffusionmodules.openaimodel.ResBlock": "supir_amd.modules.openaimodel.ResBlock",
    "sgm.modules.attention.SpatialTransformer": "supir_amd.modules.attention.SpatialTransformer",
    "sgm.modules.attention.BasicTransformerBlock": "supir_amd.modules.attention.BasicTransformerBlock",
    "sgm.modules.attention.CrossAttention": "supir_amd.modules.attention.CrossAttention",
    "sgm.modules.attention.MemoryEfficientCrossAttention": "supir_amd.modules.attention.MemoryEfficientCrossAttention",
    "SUPIR.utils.tilevae.VAEHook": "supir_amd.utils.tilevae.VAEHook",
    "sgm.modules.diffusionmodules.model.Encoder": "supir_amd.modules.vae.Encoder",
    "sgm.modules.diffusionmodules.model.Decoder": "supir_amd.modules.vae.Decoder",
    "sgm.models.autoencoder.AutoencoderKL": "supir_amd.modules.vae.AutoencoderKL",
    "sgm.models.autoencoder.AutoencoderKLInferenceWrapper": "supir_amd.modules.vae.AutoencoderKLInferenceWrapper",
    "sgm.modules.GeneralConditionerWithControl": "supir_amd.modules.conditioner.GeneralConditionerWithControl",
    "sgm.modules.GeneralConditioner": "supir_amd.modules.conditioner.GeneralConditioner",
    "sgm.modules.encoders.modules.GeneralConditionerWithControl": "supir_amd.modules.conditioner.GeneralConditionerWithControl",
    "sgm.modules.encoders.modules.FrozenCLIPEmbedder": "supir_amd.modules.conditioner.FrozenCLIPEmbedder",
    "sgm.modules.encoders.modules.FrozenOpenCLIPEmbedder2": "supir_amd.modules.conditioner.FrozenOpenCLIPEmbedder2",
    "sgm.modules.encoders.modules.ConcatTimestepEmbedderND": "supir_amd.modules.conditioner.ConcatTimestepEmbedderND",
    "SUPIR.util.PIL2Tensor": "supir_amd.utils.imageio.PIL2Tensor",
    "SUPIR.util.Tensor2PIL": "supir_amd.utils.imageio.Tensor2PIL",
    "torch.nn.Identity": "torch.nn.Identity",
}


def get_obj_from_str(string):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def _params(config):
    p = config.get("params", None) if hasattr(config, "get") else None
    return dict(p) if p is not None else {}


def instantiate_from_config(config):
    """Same contract as sgm/util.py:168-175 (`target` dotted path + `params`)."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**_params(config))


def install():
    """Register alias modules `sgm....` / `SUPIR....` in sys.modules that expose this package's classes under the
    reference's dotted paths (only for paths not already importable). Returns the list of aliased targets."""
    done = []
    for ref_path, ours in TARGET_MAP.items():
        if ref_path.startswith("torch."):
            continue
        mod_name, cls_name = ref_path.rsplit(".", 1)
        obj = get_obj_from_str(ours)
        parts = mod_name.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in sys.modules:
                try:  # a reference checkout on sys.path: patch its real module instead of shadowing it
                    importlib.import_module(name)
                except Exception:
                    m = types.ModuleType(name)
                    m.__path__ = []  # mark as package so sub-imports resolve through sys.modules
                    sys.modules[name] = m
                    if i > 1:
                        setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
        setattr(sys.modules[mod_name], cls_name, obj)
        done.append(ref_path)
    util = sys.modules.get("sgm.util")
    if util is None:
        try:
            util = importlib.import_module("sgm.util")
        except Exception:
            util = types.ModuleType("sgm.util")
            sys.modules["sgm.util"] = util
            setattr(sys.modules["sgm"], "util", util)
    # the reference resolves `target:` strings through sgm.util.get_obj_from_str (sgm/util.py:178-185): route it through
    # TARGET_MAP so that modules which imported the function by name before install() are covered too
    util.get_obj_from_str = get_obj_from_str
    if not hasattr(util, "instantiate_from_config") or getattr(util.instantiate_from_config, "__module__", "") != __name__:
        util.instantiate_from_config = instantiate_from_config
    return done
