"""SUPIRModel: the unit of work of the metric (`batchify_sample`), same constructor kwargs / methods / attribute protocol
as SUPIR/models/SUPIR_model.py:12-179 (+ the DiffusionEngine constructor subset it relies on, sgm/models/diffusion.py:23-83).

The text conditioner (CLIP-L + OpenCLIP-bigG, once per image) is built from `conditioner_config` like the
reference does (supir_amd/modules/conditioner.py: both towers on the HIP kernels).  `conditioner_config` may also be None:
`batchify_sample(..., cond=(c, uc))` takes prepared `crossattn [N,77,2048]` / `vector [N,2816]` tensors directly (what bench.py
does: synthetic conditioning, no tokeniser vocabulary files in this image), or a `conditioner` object with the reference's
`get_unconditional_conditioning(batch, batch_uc)` method can be attached.
"""
import contextlib
import copy
import random

import torch
import torch.nn as nn

from .. import weights as Wt
from ..modules.vae import DiagonalGaussianDistribution
from ..plugin import get_obj_from_str, instantiate_from_config

_DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def seed_everything(seed):
    """pytorch_lightning.seed_everything: python / numpy / torch global RNGs (SUPIR_model.py:115)."""
    import numpy as np
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    return seed


def _cfg_set(cfg, path, value):
    node = cfg
    for k in path[:-1]:
        node = node[k]
    node[path[-1]] = value


class SUPIRModel(nn.Module):
    def __init__(self, control_stage_config, network_config, denoiser_config, first_stage_config, sampler_config,
                 conditioner_config=None, ae_dtype="fp32", diffusion_dtype="fp32", p_p="", n_p="", scale_factor=1.0,
                 disable_first_stage_autocast=False, network_wrapper=None, **kwargs):
        super().__init__()
        net = instantiate_from_config(network_config)
        wrapper = get_obj_from_str(network_wrapper or "sgm.modules.diffusionmodules.wrappers.ControlWrapper")
        self.model = wrapper(net)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(conditioner_config) if conditioner_config is not None else None
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        self.scale_factor = scale_factor
        self.model.load_control_model(instantiate_from_config(control_stage_config))
        self.first_stage_model.denoise_encoder = copy.deepcopy(self.first_stage_model.encoder)
        self.sampler_config = copy.deepcopy(sampler_config)
        assert ae_dtype in _DT and diffusion_dtype in _DT
        if ae_dtype == "fp16":
            raise RuntimeError("fp16 cause NaN in AE")  # SUPIR_model.py:23-24
        self.ae_dtype = _DT[ae_dtype]
        self.model.dtype = _DT[diffusion_dtype]
        self.p_p, self.n_p = p_p, n_p
        self._ae_dtype_noted = None

    # ------------------------------------------------------------------ first stage
    @contextlib.contextmanager
    def _ae_scope(self):
        """The reference runs the three VAE entry points under `torch.autocast("cuda", dtype=self.ae_dtype)` (SUPIR_model.py:41-69;
        `model.ae_dtype = ...` is part of test.py's attribute protocol, test.py:67).  Here `ae_dtype` selects the element type of
        the VAE kernels for the call: torch.bfloat16 (test.py's default) -> bf16, the reference's own arithmetic for that request.
        torch.float32 (`--ae_dtype fp32`) is computed by the reference in TRUE fp32 (autocast disables itself for float32) -> the fp32
        service (libsupir_hip_f32.so, supir_amd/ops_f32.py); with weights.FP32_NATIVE off (SUPIR_FP32_NATIVE=0) it is served in bf16 --
        said once per request with a RuntimeWarning, a RuntimeError under SUPIR_STRICT_DTYPE=1.  torch.float16 is refused like the
        reference's constructor refuses it (SUPIR_model.py:23-24)."""
        dt = self.ae_dtype
        if dt == torch.float16:
            raise RuntimeError("fp16 cause NaN in AE")
        served = Wt.as_compute_dtype(dt)
        if served != dt and dt != torch.bfloat16 and self._ae_dtype_noted != dt:
            self._ae_dtype_noted = dt
            Wt.note_downgrade("SUPIRModel.ae_dtype", f"{dt} (test.py --ae_dtype fp32)", "torch.bfloat16",
                              "the reference computes this request in true fp32 (torch.autocast disables itself for float32, "
                              "SUPIR/models/SUPIR_model.py:41-69); SUPIR_FP32_NATIVE=0 keeps fp32 requests off the fp32 service "
                              "(decoder rel-L2 vs fp32 ~1e-2 at 512 px).", stacklevel=5)
        with Wt.compute_dtype(served):
            yield

    @torch.no_grad()
    def encode_first_stage(self, x, noise=None):
        fs = self.first_stage_model
        with self._ae_scope():
            post = DiagonalGaussianDistribution(fs.quant_conv(fs.encoder(x)))
            return self.scale_factor * post.sample(noise)

    @torch.no_grad()
    def encode_first_stage_with_denoise(self, x, use_sample=True, is_stage1=False, noise=None):
        fs = self.first_stage_model
        with self._ae_scope():
            h = fs.denoise_encoder_s1(x) if is_stage1 else fs.denoise_encoder(x)
            post = DiagonalGaussianDistribution(fs.quant_conv(h))
            z = post.sample(noise) if use_sample else post.mode()
            return self.scale_factor * z

    @torch.no_grad()
    def decode_first_stage(self, z):
        fs = self.first_stage_model
        with self._ae_scope():
            return fs.decoder(fs.post_quant_conv(z, in_scale=1.0 / self.scale_factor)).float()

    @torch.no_grad()
    def batchify_denoise(self, x, is_stage1=False):
        return self.decode_first_stage(self.encode_first_stage_with_denoise(x, use_sample=False, is_stage1=is_stage1))

    # ------------------------------------------------------------------ the unit of work
    @torch.no_grad()
    def batchify_sample(self, x, p=None, p_p="default", n_p="default", num_steps=100, restoration_scale=4.0, s_churn=0,
                        s_noise=1.003, cfg_scale=4.0, seed=-1, num_samples=1, control_scale=1, color_fix_type="None",
                        use_linear_CFG=False, use_linear_control_scale=False, cfg_scale_start=1.0,
                        control_scale_start=0.0, cond=None, noises=None, return_intermediates=False, **kwargs):
        """SUPIR_model.py:80-136.  `cond=(c, uc)` bypasses the text conditioner; `noises` (dict with optional
        'posterior', 'init', 'steps') injects the RNG draws for parity runs (otherwise the sampler's churn noise comes from
        torch.randn_like on the device generator exactly like the reference)."""
        assert color_fix_type in ["Wavelet", "AdaIn", "None"]
        N = len(x)
        if num_samples > 1:
            assert N == 1
            N = num_samples
            x = x.repeat(N, 1, 1, 1)
            p = p * N if p is not None else None
        p_p = self.p_p if p_p == "default" else p_p
        n_p = self.n_p if n_p == "default" else n_p
        sc = self.sampler_config
        _cfg_set(sc, ("params", "num_steps"), num_steps)
        _cfg_set(sc, ("params", "guider_config", "params", "scale_min"), cfg_scale)
        _cfg_set(sc, ("params", "guider_config", "params", "scale"), cfg_scale_start if use_linear_CFG else cfg_scale)
        _cfg_set(sc, ("params", "restore_cfg"), restoration_scale)
        _cfg_set(sc, ("params", "s_churn"), s_churn)
        _cfg_set(sc, ("params", "s_noise"), s_noise)
        self.sampler = instantiate_from_config(sc)
        if seed == -1:
            seed = random.randint(0, 65535)
        seed_everything(seed)
        noises = noises or {}
        _z = self.encode_first_stage_with_denoise(x, use_sample=False)
        x_stage1 = self.decode_first_stage(_z)
        z_stage1 = self.encode_first_stage(x_stage1, noise=noises.get("posterior"))
        if cond is None:
            assert len(x) == len(p)   # SUPIR_model.py:95
        if cond is not None:
            c, uc = dict(cond[0]), dict(cond[1])
            c["control"] = _z
            uc["control"] = _z
        else:
            c, uc = self.prepare_condition(_z, p, p_p, n_p, N)
        denoiser = lambda inp, sigma, cc, cs: self.denoiser(self.model, inp, sigma, cc, cs, **kwargs)
        if not kwargs:   # lets RestoreEDMSampler fuse the elementwise halves of a step around the network call (sampling.py)
            denoiser.fused = (self.denoiser, self.model)
        noised_z = noises["init"].to(_z).clone() if "init" in noises else torch.randn_like(_z)
        if "steps" in noises:   # parity runs: per-step churn noise instead of torch.randn_like (RestoreEDMSampler only)
            self.sampler.injected_step_noises = list(noises["steps"])
        _samples = self.sampler(denoiser, noised_z, cond=c, uc=uc, x_center=z_stage1, control_scale=control_scale,
                                use_linear_control_scale=use_linear_control_scale, control_scale_start=control_scale_start)
        samples = self.decode_first_stage(_samples)
        if color_fix_type == "Wavelet":
            from ..utils.colorfix import wavelet_reconstruction
            samples = wavelet_reconstruction(samples, x_stage1)
        elif color_fix_type == "AdaIn":
            from ..utils.colorfix import adaptive_instance_normalization
            samples = adaptive_instance_normalization(samples, x_stage1)
        if return_intermediates:
            return samples, dict(z=_z, x_stage1=x_stage1, z_stage1=z_stage1, samples=_samples)
        return samples

    def init_tile_vae(self, encoder_tile_size=512, decoder_tile_size=64):
        """SUPIR_model.py:138-150: route the three VAE nets through the tiled forward (test.py --use_tile_vae)."""
        from ..utils.tilevae import VAEHook
        fs = self.first_stage_model
        for net, size, dec in ((fs.denoise_encoder, encoder_tile_size, False), (fs.encoder, encoder_tile_size, False),
                               (fs.decoder, decoder_tile_size, True)):
            net.original_forward = net.forward
            net.forward = VAEHook(net, size, is_decoder=dec, fast_decoder=False, fast_encoder=False, color_fix=False,
                                  to_gpu=True)

    def prepare_condition(self, _z, p, p_p, n_p, N):
        if self.conditioner is None:
            raise RuntimeError("no text conditioner attached (conditioner_config was None); pass cond=(c, uc) with "
                               "crossattn [N,77,2048] / vector [N,2816], or attach a conditioner object")
        batch = {"original_size_as_tuple": torch.tensor([1024, 1024]).repeat(N, 1).to(_z.device),
                 "crop_coords_top_left": torch.tensor([0, 0]).repeat(N, 1).to(_z.device),
                 "target_size_as_tuple": torch.tensor([1024, 1024]).repeat(N, 1).to(_z.device),
                 "aesthetic_score": torch.tensor([9.0]).repeat(N, 1).to(_z.device), "control": _z}
        batch_uc = copy.deepcopy(batch)
        batch_uc["txt"] = [n_p for _ in p]
        # the reference runs the conditioner under autocast(ae_dtype) (SUPIR_model.py:165,174): the text towers follow the VAE's request --
        # bf16 by test.py's default, true fp32 for `--ae_dtype fp32`
        if not isinstance(p[0], list):
            batch["txt"] = ["".join([_p, p_p]) for _p in p]
            with self._ae_scope():
                return self.conditioner.get_unconditional_conditioning(batch, batch_uc)
        # local (per-tile) prompts, SUPIR_model.py:168-178: one cond per tile, a single uc (the tiled samplers' `cond` list)
        assert len(p) == 1, "Support bs=1 only for local prompt conditioning."
        c, uc = [], None
        for i, p_tile in enumerate(p[0]):
            batch["txt"] = ["".join([p_tile, p_p])]
            with self._ae_scope():
                if i == 0:
                    _c, uc = self.conditioner.get_unconditional_conditioning(batch, batch_uc)
                else:
                    _c, _ = self.conditioner.get_unconditional_conditioning(batch, None)
            c.append(_c)
        return c, uc
