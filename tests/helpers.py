"""Shared test helpers: synthetic state dicts from the committed manifests, golden loading, error metrics."""
import json
import os

import torch

from supir_amd.synth import synth_param, synth_tensor  # noqa: F401

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest(name="mini"):
    return json.load(open(os.path.join(GOLDEN_DIR, f"manifest_{name}.json")))


def synth_sd(man, device="cpu", prefixes=None):
    sd = {}
    for k, shape in man.items():
        if prefixes is None or any(k.startswith(p) for p in prefixes):
            sd[k] = synth_param(k, shape, device=device)
    return sd


_gold = {}


def golden():
    if "g" not in _gold:
        _gold["g"] = torch.load(os.path.join(GOLDEN_DIR, "golden_mini.pt"), map_location="cpu", weights_only=False)
    return _gold["g"]


def golden_control():
    """The ten GLVControl feature maps of the reference as full tensors (oracle/gen_golden_control.py)."""
    if "c" not in _gold:
        _gold["c"] = torch.load(os.path.join(GOLDEN_DIR, "golden_control.pt"), map_location="cpu", weights_only=False)["control_features"]
    return _gold["c"]


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


# ------------------------------------------------------------------------------------------ product-model builders
SUPIR_NET = dict(mode="XL-base", project_type="ZeroSFT", project_channel_scale=2, adm_in_channels=2816,
                 num_classes="sequential", use_checkpoint=True, in_channels=4, out_channels=4, model_channels=320,
                 attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64,
                 use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=[1, 2, 10],
                 context_dim=2048, spatial_transformer_attn_type="softmax-xformers", legacy=False)
VAE_DD = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def fill_module(module, prefix, device):
    """Synthetic weights by reference key name (prefix + local name), generated directly on `device`."""
    with torch.no_grad():
        for k, t in module.state_dict().items():
            if t.is_floating_point() and not k.endswith("sigmas"):
                t.copy_(synth_param(prefix + k, t.shape, device=device))
    return module


def build_unet(depth=(1, 2, 10), device="cuda"):
    from supir_amd.modules.supir_v0 import GLVControl, LightGLVUNet
    from supir_amd.modules.wrappers import ControlWrapper
    net = dict(SUPIR_NET, transformer_depth=list(depth))
    ctl = {k: v for k, v in net.items() if k not in ("mode", "project_type", "project_channel_scale")}
    with torch.device(device):
        unet = LightGLVUNet(**net)
        ctrl = GLVControl(**ctl, input_upscale=1)
    fill_module(unet, "model.diffusion_model.", device)
    fill_module(ctrl, "model.control_model.", device)
    wrap = ControlWrapper(unet, dtype=torch.bfloat16)
    wrap.load_control_model(ctrl)
    return wrap


def build_vae(device="cuda"):
    import copy
    from supir_amd.modules.vae import AutoencoderKLInferenceWrapper
    with torch.device(device):
        vae = AutoencoderKLInferenceWrapper(embed_dim=4, ddconfig=dict(VAE_DD), lossconfig={"target": "torch.nn.Identity"})
        vae.denoise_encoder = copy.deepcopy(vae.encoder)
    fill_module(vae, "first_stage_model.", device)
    return vae
