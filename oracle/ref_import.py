"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Imports the REAL reference (/root/reference, read-only, present only in the build container) on CPU so that
(a) the restatement in supir_oracle.py can be validated against it and (b) golden vectors can be generated
(gen_golden.py).  The reference needs pip packages that are not installed here (omegaconf, pytorch_lightning,
k_diffusion, kornia, open_clip, cv2, diffusers, torchvision, xformers); they are replaced by inert in-memory stubs
(recipe: SURVEY.md section 8(c)).  Nothing here is copied from the reference; nothing here runs on the GPU box.
"""
import contextlib
import io
import os
import sys
import types

REF_ROOT = os.environ.get("SUPIR_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "sgm"))


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return _AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return _ListConfig(_wrap(v) for v in o)
    return o


class _ListConfig(list):
    pass


def _install_stubs():
    import torch
    import torch.nn as nn
    import yaml

    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    # transformers' CLIP classes must be imported BEFORE a fake torchvision exists
    try:
        from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    except Exception:
        pass

    class OmegaConf:
        @staticmethod
        def load(path):
            with open(path) as f:
                return _wrap(yaml.safe_load(f))

        @staticmethod
        def create(o):
            return _wrap(o)

        @staticmethod
        def to_container(o, **kw):
            return o

    oc = mod("omegaconf", OmegaConf=OmegaConf, ListConfig=_ListConfig, DictConfig=_AttrDict)
    mod("omegaconf.listconfig", ListConfig=_ListConfig)
    oc.listconfig = sys.modules["omegaconf.listconfig"]

    class LightningModule(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    def seed_everything(seed):
        import random

        import numpy as np
        random.seed(seed)
        np.random.seed(seed % (2 ** 32))
        torch.manual_seed(seed)
        return seed

    mod("pytorch_lightning", LightningModule=LightningModule, seed_everything=seed_everything)
    mod("k_diffusion")
    mod("k_diffusion.sampling", get_sigmas_karras=None, BrownianTreeNoiseSampler=None)
    mod("kornia")
    mod("open_clip")
    mod("cv2")
    mod("diffusers")
    mod("diffusers.utils")
    mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.transforms = mod("torchvision.transforms", ToTensor=object, ToPILImage=object)
    tv.__version__ = "0.0"


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's hot-path classes. Constructor prints are swallowed."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference not mounted")
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with contextlib.redirect_stdout(io.StringIO()):
        from SUPIR.modules.SUPIR_v0 import GLVControl, LightGLVUNet, ZeroCrossAttn, ZeroSFT
        from sgm.modules.attention import BasicTransformerBlock, SpatialTransformer
        from sgm.modules.diffusionmodules import sampling
        from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiserWithControl
        from sgm.modules.diffusionmodules.guiders import LinearCFG
        from sgm.modules.diffusionmodules.model import Decoder, Encoder
        from sgm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
        from sgm.modules.diffusionmodules.wrappers import ControlWrapper
        from sgm.modules.distributions.distributions import DiagonalGaussianDistribution
        from sgm.util import instantiate_from_config
        from omegaconf import OmegaConf
    ns = types.SimpleNamespace(**{k: v for k, v in locals().items() if not k.startswith("_")})
    ns.root = REF_ROOT
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def unet_params(depth=None):
    """network_config / control_stage_config params of options/SUPIR_v0.yaml (optionally reduced transformer depth)."""
    ns = load_reference()
    cfg = ns.OmegaConf.load(os.path.join(REF_ROOT, "options", "SUPIR_v0.yaml"))
    net = dict(cfg.model.params.network_config.params)
    ctl = dict(cfg.model.params.control_stage_config.params)
    for d in (net, ctl):
        d["spatial_transformer_attn_type"] = "softmax"  # no xformers here: the reference's own SDPA fallback
        if depth is not None:
            d["transformer_depth"] = list(depth)
    vae = dict(cfg.model.params.first_stage_config.params.ddconfig)
    vae["attn_type"] = "vanilla"
    return net, ctl, vae
