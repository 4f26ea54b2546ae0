"""Host-side numerics of the sampling loop: discretisation, denoiser scaling, CFG guider, Restore-EDM samplers.

Same classes / constructor kwargs / call signatures as sgm/modules/diffusionmodules/{discretizer.py:42-69,
denoiser.py:31-73, denoiser_scaling.py:16-22, guiders.py:44-74, sampling.py:25-61,528-660,733-766}.  These are scalar
schedules and elementwise updates on [N,4,h,w] latents ("negligible bytes; keep in PyTorch"): they stay
torch ops on the device; what changes is that the sigma schedule lives on the HOST, so the loop has no device->host
sync per step (the reference compares device tensors in Python twice per step, sampling.py:563,581).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .brownian import BrownianTreeNoiseSampler

SIGMA_MAX = 14.6146
# RestoreEDMSampler: run the elementwise halves of a step as two fused kernels (csrc/sampler.hip) with host-side scalars when the
# caller exposes its denoiser / network (SUPIRModel.batchify_sample does); off -> the generic torch-op path for every caller
FUSED_EDM_STEP = os.environ.get("SUPIR_FUSED_EDM_STEP", "1") == "1"
# with the fused step: the network prepares the time / label embeddings of ALL steps of an image before the loop
# (ControlWrapper.prepare_schedule) instead of recomputing them at the head of every step
EMB_SCHEDULE = os.environ.get("SUPIR_EMB_SCHEDULE", "1") == "1"


def append_dims(x, ndim):
    return x[(...,) + (None,) * (ndim - x.ndim)]


# ----------------------------------------------------------------------------------------------- discretisation
class LegacyDDPMDiscretization:
    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            ts = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
            ac = self.alphas_cumprod[ts]
        elif n == self.num_timesteps:
            ac = self.alphas_cumprod
        else:
            raise ValueError
        sig = torch.tensor((1 - ac) / ac, dtype=torch.float32, device=device) ** 0.5
        return torch.flip(sig, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sig = self.get_sigmas(n, device=device)
        if do_append_zero:
            sig = torch.cat([sig, sig.new_zeros([1])])
        return sig if not flip else torch.flip(sig, (0,))


class EpsScaling:
    def __call__(self, sigma):
        return torch.ones_like(sigma), -sigma, 1 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class EpsWeighting:
    def __call__(self, sigma):
        return sigma ** -2.0


def _make(cfg, default):
    """Accepts None, an instance, or a reference-style {"target": ..., "params": ...} config."""
    if cfg is None:
        return default()
    if isinstance(cfg, dict) or hasattr(cfg, "get") and cfg.get("target") is not None:
        from ..plugin import instantiate_from_config
        return instantiate_from_config(cfg)
    return cfg


# ----------------------------------------------------------------------------------------------- denoiser
class DiscreteDenoiserWithControl(nn.Module):
    def __init__(self, weighting_config=None, scaling_config=None, num_idx=1000, discretization_config=None,
                 do_append_zero=False, quantize_c_noise=True, flip=True):
        super().__init__()
        self.weighting = _make(weighting_config, EpsWeighting)
        self.scaling = _make(scaling_config, EpsScaling)
        sigmas = _make(discretization_config, LegacyDDPMDiscretization)(num_idx, do_append_zero=do_append_zero, flip=flip)
        self.register_buffer("sigmas", sigmas)
        self.quantize_c_noise = quantize_c_noise

    def _table(self, like):
        if self.sigmas.device != like.device:  # built under a device context / before .to(): follow the data
            self.sigmas = self.sigmas.to(like.device)
        return self.sigmas

    def sigma_to_idx(self, sigma):
        return (sigma - self._table(sigma)[:, None]).abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self._table(idx)[idx]

    def w(self, sigma):
        return self.weighting(sigma)

    # ---- host-side mirror of __call__'s scalar arithmetic for a batch-uniform sigma (the fused sampler step)
    def host_scalars(self, sigma32):
        """sigma (numpy float32 scalar) -> (table index, c_skip, c_out, c_in) exactly as __call__ computes them on the device in
        fp32: snap to the nearest table entry (first minimum, like argmin), EpsScaling in the reference's operation order
        (denoiser.py:49-73, denoiser_scaling.py:16-22).  None when the configuration is not the one SUPIR ships."""
        if not (isinstance(self.scaling, EpsScaling) and self.quantize_c_noise):
            return None
        if self.sigmas.dtype != torch.float32:
            # `model.half()` (test.py --loading_half_params) halves this buffer as it does the reference's: __call__ then does its
            # scalar arithmetic in fp16 like the reference; the fp32 host mirror would not reproduce that -> generic path
            return None
        key = (self.sigmas.data_ptr(), self.sigmas._version, self.sigmas.device)   # `denoiser.sigmas` is a checkpoint key
        cached = self.__dict__.get("_host_table")
        if cached is None or cached[0] != key:
            cached = (key, self.sigmas.detach().float().cpu().numpy().astype(np.float32))
            self.__dict__["_host_table"] = cached
        tab = cached[1]
        idx = int(np.argmin(np.abs(np.float32(sigma32) - tab)))
        sq = np.float32(tab[idx])
        c_in = np.float32(1.0) / np.sqrt(sq * sq + np.float32(1.0), dtype=np.float32)
        return idx, 1.0, float(-sq), float(np.float32(c_in))

    def idx_tensor(self, idx, count, device):
        """int64 [count] filled with a table index (what the network receives as c_noise); cached: a 50-step schedule visits the
        same 50 indices for every image."""
        cache = self.__dict__.setdefault("_idx_cache", {})
        key = (idx, count, str(device))
        t = cache.get(key)
        if t is None:
            if len(cache) > 4096:
                cache.clear()
            t = torch.full((count,), idx, dtype=torch.int64, device=device)
            cache[key] = t
        return t

    def __call__(self, network, input, sigma, cond, control_scale, **kwargs):
        """denoiser.py:66-73: sigma snapped to the 1000-entry table; the network sees the int64 table index."""
        idx = self.sigma_to_idx(sigma)
        sigma = self.idx_to_sigma(idx)
        s = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(s)
        c_noise = self.sigma_to_idx(c_noise.reshape(sigma.shape)) if self.quantize_c_noise else c_noise.reshape(sigma.shape)
        return network(input * c_in, c_noise, cond, control_scale, **kwargs) * c_out + input * c_skip


# ----------------------------------------------------------------------------------------------- guiders
class NoDynamicThresholding:
    def __call__(self, uncond, cond, scale):
        return uncond + scale.view(-1, 1, 1, 1) * (cond - uncond)


_CAT_KEYS = ("vector", "crossattn", "concat", "control", "control_vector", "mask_x")


class LinearCFG:
    def __init__(self, scale, scale_min=None, dyn_thresh_config=None):
        self.scale, self.scale_min = scale, scale if scale_min is None else scale_min
        self.dyn_thresh = NoDynamicThresholding()

    def scale_schedule(self, sigma):
        return (self.scale - self.scale_min) * sigma / SIGMA_MAX + self.scale_min

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return self.dyn_thresh(x_u, x_c, self.scale_schedule(sigma))

    def prepare_cond(self, c, uc):
        """[uncond; cond] concat of the conditioning (guiders.py:65-74). Step independent, so samplers call it ONCE."""
        out = {}
        for k in c:
            if k in _CAT_KEYS:
                out[k] = torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] == uc[k]
                out[k] = c[k]
        return out

    def prepare_inputs(self, x, s, c, uc):
        return torch.cat([x] * 2), torch.cat([s] * 2), self.prepare_cond(c, uc)


class VanillaCFG(LinearCFG):
    def __init__(self, scale, dyn_thresh_config=None):
        super().__init__(scale, scale)


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_cond(self, c, uc):
        return dict(c)

    def prepare_inputs(self, x, s, c, uc):
        return x, s, dict(c)


# ----------------------------------------------------------------------------------------------- samplers
class BaseDiffusionSampler:
    def __init__(self, discretization_config=None, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        self.num_steps = num_steps
        self.discretization = _make(discretization_config, LegacyDDPMDiscretization)
        self.guider = _make(guider_config, IdentityGuider)
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        n = self.num_steps if num_steps is None else num_steps
        sig_host = self.discretization(n, device="cpu")          # schedule on the host: no per-step syncs
        sigmas = sig_host.to(x.device)
        uc = cond if uc is None else uc
        x *= torch.sqrt(1.0 + sigmas[0] ** 2.0)                  # in place, like the reference (sampling.py:51)
        s_in = x.new_ones([x.shape[0]])
        return x, s_in, sigmas, len(sigmas), cond, uc, [float(v) for v in sig_host]


class RestoreEDMSampler(BaseDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, restore_cfg=4.0,
                 restore_cfg_s_tmin=0.05, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        self.restore_cfg, self.restore_cfg_s_tmin = restore_cfg, restore_cfg_s_tmin
        self.sigma_max = SIGMA_MAX

    def denoise(self, x, denoiser, sigma, cond, uc, control_scale=1.0, cond_cat=None):
        if cond_cat is None:
            cond_cat = self.guider.prepare_cond(cond, uc)
        twice = not isinstance(self.guider, IdentityGuider)
        xin = torch.cat([x] * 2) if twice else x
        sin = torch.cat([sigma] * 2) if twice else sigma
        return self.guider(denoiser(xin, sin, cond_cat, control_scale), sigma)

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0, x_center=None, eps_noise=None,
                     control_scale=1.0, use_linear_control_scale=False, control_scale_start=0.0, cond_cat=None,
                     sigma_f=None, next_sigma_f=None):
        """sampling.py:548-570. sigma_f / next_sigma_f: host copies of sigma[0] / next_sigma[0] (avoid device syncs)."""
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            eps = (eps_noise if eps_noise is not None else torch.randn_like(x)) * self.s_noise
            x = x + eps * append_dims(sigma_hat ** 2 - sigma ** 2, x.ndim) ** 0.5
        if sigma_f is None:
            sigma_f, next_sigma_f = float(sigma[0]), float(next_sigma[0])
        if use_linear_control_scale:
            control_scale = (sigma_f / self.sigma_max) * (control_scale_start - control_scale) + control_scale
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc, control_scale=control_scale, cond_cat=cond_cat)
        if (next_sigma_f > self.restore_cfg_s_tmin) and (self.restore_cfg > 0):
            d_center = denoised - x_center
            denoised = denoised - d_center * ((sigma.view(-1, 1, 1, 1) / self.sigma_max) ** self.restore_cfg)
        d = (x - denoised) / append_dims(sigma_hat, x.ndim)
        return x + append_dims(next_sigma - sigma_hat, x.ndim) * d

    def _gamma(self, sig_f, num_sigmas):
        return min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sig_f <= self.s_tmax else 0.0

    # ------------------------------------------------------------------ fused step (two elementwise kernels + the network call)
    def _fused_ctx(self, denoiser, x):
        """(denoiser module, network) when this call can take the fused step: fp32 CUDA latents, a caller that exposes what its
        denoiser closure wraps (`denoiser.fused = (DiscreteDenoiserWithControl, network)`), the guiders SUPIR ships."""
        f = getattr(denoiser, "fused", None)
        if not FUSED_EDM_STEP or f is None or not x.is_cuda or x.dtype != torch.float32:
            return None
        if not isinstance(self.guider, (LinearCFG, IdentityGuider)):
            return None
        if isinstance(self.guider, LinearCFG) and not isinstance(self.guider.dyn_thresh, NoDynamicThresholding):
            return None
        den, net = f
        if not isinstance(den, DiscreteDenoiserWithControl) or den.host_scalars(np.float32(1.0)) is None:
            return None
        return den, net

    def _fused_step(self, ctx, sigma_f, next_sigma_f, x, gamma, x_center, eps_noise, control_scale, use_linear_control_scale,
                    control_scale_start, cond_cat, sched_row=None, tiles=None):
        """sampler_step (sampling.py:548-570) with every sigma-derived factor evaluated on the host in fp32, in the reference's
        operation order, and the tensor work in supir_edm_step_pre / _post.  Same RNG consumption as the generic path.
        tiles = (windows, T): x / eps_noise are whole canvases and the step runs on the k stacked T x T windows of them
        (supir_edm_step_pre_tiles does the crop); x_center is then the pre-stacked centre [k*b, C, T, T] and the result the stacked
        x_next of the k tiles."""
        from .. import ops
        den, net = ctx
        f32 = np.float32
        sigma, nxt = f32(sigma_f), f32(next_sigma_f)
        sigma_hat = sigma * f32(gamma + 1.0)
        eps, noise_mul = None, 0.0
        if gamma > 0:
            eps = eps_noise if eps_noise is not None else torch.randn_like(x)
            noise_mul = float(np.sqrt(sigma_hat * sigma_hat - sigma * sigma, dtype=f32))
        if use_linear_control_scale:
            control_scale = (sigma_f / self.sigma_max) * (control_scale_start - control_scale) + control_scale
        idx, c_skip, c_out, c_in = den.host_scalars(sigma_hat)
        twice = not isinstance(self.guider, IdentityGuider)
        reps = 2 if twice else 1
        x = x if x.is_contiguous() else x.contiguous()
        if tiles is not None:
            x_hat, net_in = ops.edm_step_pre_tiles(x, None if eps is None else eps.float().contiguous(), tiles[0], tiles[1],
                                                   self.s_noise, noise_mul, c_in, reps)
        else:
            x_hat, net_in = ops.edm_step_pre(x, None if eps is None else eps.float().contiguous(), self.s_noise, noise_mul, c_in, reps)
        if sched_row is not None:
            net.select_step(sched_row, idx)     # this call's timestep embeddings come out of the per-image table (ControlWrapper.prepare_schedule)
        out = net(net_in, den.idx_tensor(idx, net_in.shape[0], x.device), cond_cat, control_scale)
        cfg = 0.0
        if twice:
            g = self.guider
            cfg = float(f32(g.scale - g.scale_min) * sigma_hat / f32(SIGMA_MAX) + f32(g.scale_min))
        center, restore_mul = None, 0.0
        if (next_sigma_f > self.restore_cfg_s_tmin) and (self.restore_cfg > 0):
            center = x_center
            restore_mul = float(np.power(sigma / f32(self.sigma_max), f32(self.restore_cfg), dtype=f32))
        return ops.edm_step_post(out.float().contiguous(), x_hat, center, c_out, c_skip, cfg, restore_mul, float(sigma_hat),
                                 float(nxt - sigma_hat), reps)

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, x_center=None, control_scale=1.0,
                 use_linear_control_scale=False, control_scale_start=0.0):
        x, s_in, sigmas, num_sigmas, cond, uc, sf = self.prepare_sampling_loop(x, cond, uc, num_steps)
        cond_cat = self.guider.prepare_cond(cond, uc)            # constant over the loop: concat once
        inject = self.__dict__.pop("injected_step_noises", None)  # parity runs: the churn noise of every step, given
        ctx = self._fused_ctx(denoiser, x) if type(self).sampler_step is RestoreEDMSampler.sampler_step else None
        sched = False
        if ctx is not None and EMB_SCHEDULE and hasattr(ctx[1], "prepare_schedule") and "vector" in cond_cat:
            # every step's timestep (table index) is known now: let the network prepare all its time / label embeddings at once
            f32 = np.float32
            t_all = [ctx[0].host_scalars(f32(sf[i]) * f32(self._gamma(sf[i], num_sigmas) + 1.0))[0] for i in range(num_sigmas - 1)]
            ctx[1].prepare_schedule(t_all, cond_cat["vector"], control=cond_cat.get("control"))
            sched = True
        try:
            for i in range(num_sigmas - 1):
                if ctx is None:
                    break
                x = self._fused_step(ctx, sf[i], sf[i + 1], x, self._gamma(sf[i], num_sigmas), x_center,
                                     None if inject is None else inject[i].to(x), control_scale, use_linear_control_scale,
                                     control_scale_start, cond_cat, sched_row=i if sched else None)
        finally:
            if sched:
                ctx[1].end_schedule()
        if ctx is not None:
            return x
        for i in range(num_sigmas - 1):
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, self._gamma(sf[i], num_sigmas),
                                  x_center, eps_noise=None if inject is None else inject[i].to(x), control_scale=control_scale, use_linear_control_scale=use_linear_control_scale,
                                  control_scale_start=control_scale_start, cond_cat=cond_cat, sigma_f=sf[i],
                                  next_sigma_f=sf[i + 1])
        return x


def gaussian_weights(tile_width, tile_height, nbatches, device="cuda"):
    """sampling.py:733-750 (float64; x midpoint (w-1)/2, y midpoint h/2 -- reproduced as is)."""
    from numpy import exp, pi, sqrt
    var = 0.01
    mid = (tile_width - 1) / 2
    xp = [exp(-(x - mid) * (x - mid) / (tile_width * tile_width) / (2 * var)) / sqrt(2 * pi * var) for x in range(tile_width)]
    mid = tile_height / 2
    yp = [exp(-(y - mid) * (y - mid) / (tile_height * tile_height) / (2 * var)) / sqrt(2 * pi * var) for y in range(tile_height)]
    return torch.tile(torch.tensor(np.outer(yp, xp), device=device), (nbatches, 4, 1, 1))


def _sliding_windows(h, w, tile_size, tile_stride):
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


class TiledRestoreEDMSampler(RestoreEDMSampler):
    """sampling.py:600-660: per step, every 128x128 latent tile takes a full sampler step; Gaussian-weighted blend."""

    def __init__(self, tile_size=128, tile_stride=64, *args, tile_batch=1, tile_parallel=False, process_group=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.tile_size, self.tile_stride = tile_size, tile_stride
        self.tile_weights = None
        # with `tile_parallel` and an initialised torch.distributed group, the tile groups of a step are dealt
        # round-robin to the ranks (one process per GPU); every rank blends its own tiles into a zero canvas and ONE all-reduce
        # (sum, RCCL) of that canvas per step assembles x_next -- the only exchange the algorithm needs (the normalising
        # `count` canvas is tile-geometry only, computed locally).  Rank 0's start latent and per-step churn noise are broadcast
        # so the ranks cannot drift apart even if their RNG streams differ.
        self.tile_parallel = bool(tile_parallel)
        self.process_group = process_group
        # tiles of one step are independent given x and eps_noise (sampling.py:629-657): `tile_batch` > 1 stacks that many
        # tiles along the batch axis of ONE denoiser call (M of every GEMM grows k-fold, 49 -> ceil(49/k) launches of the
        # network per step); per-tile arithmetic is unchanged, so the result is identical.
        self.tile_batch = max(1, int(tile_batch))

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, x_center=None, control_scale=1.0,
                 use_linear_control_scale=False, control_scale_start=0.0):
        use_local_prompt = isinstance(cond, list)
        self.__dict__["_rep_cache"] = {}   # the repeated-conditioning cache is per call (`static` below is rebuilt every call)
        b, _, h, w = x.shape
        tiles = _sliding_windows(h, w, self.tile_size, self.tile_stride)
        if self.tile_weights is None or self.tile_weights.device != x.device:
            self.tile_weights = gaussian_weights(self.tile_size, self.tile_size, 1, device=x.device)
        tile_weights = self.tile_weights.repeat(b, 1, 1, 1)
        if use_local_prompt:
            assert len(cond) == len(tiles), "Number of local prompts should be equal to number of tiles"
            lq = cond[0]["control"]
        else:
            lq = cond["control"]
        x, s_in, sigmas, num_sigmas, cond, uc, sf = self.prepare_sampling_loop(x, cond, uc, num_steps)
        uc = dict(uc)
        conds = [dict(cj) for cj in cond] if use_local_prompt else [dict(cond)]
        # the text / vector halves of the CFG batch are tile independent: concat them once per prompt
        static = []
        for cj in conds:
            cat = self.guider.prepare_cond({k: v for k, v in cj.items() if k != "control"},
                                           {k: v for k, v in uc.items() if k != "control"})
            static.append(cat)
        import torch.distributed as dist
        from ..parallel import collectives_active
        world, rank = 1, 0
        shared = self.tile_parallel and collectives_active(self.process_group)   # > 1 rank (or a forced 1-rank group: RCCL test)
        if shared:
            world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        kb = 1 if use_local_prompt else self.tile_batch
        src = 0
        if shared:
            src = dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0
            x = x.contiguous()
            dist.broadcast(x, src=src, group=self.process_group)
        ctx = None
        # the fused step's two tile kernels stack at most 64 tiles per group (SUPIR_MAX_TILES, csrc/kernels.h) and take exact T x T windows
        # inside the canvas; anything else (tile_batch > 64, a canvas smaller than a tile) runs the generic loop below
        fits = kb <= 64 and x.shape[-2] >= self.tile_size and x.shape[-1] >= self.tile_size
        if fits and not use_local_prompt and type(self).sampler_step is RestoreEDMSampler.sampler_step and x_center is not None:
            ctx = self._fused_ctx(denoiser, x)
        if ctx is not None:
            return self._call_fused(ctx, x, tiles, tile_weights, lq, static[0], sf, num_sigmas, x_center, control_scale,
                                    use_linear_control_scale, control_scale_start, kb, shared, world, rank, src)
        if shared:
            count_all = torch.zeros_like(x)
            for (hi, he, wi, we) in tiles:
                count_all[:, :, hi:he, wi:we] += tile_weights
        for i in range(num_sigmas - 1):
            gamma = self._gamma(sf[i], num_sigmas)
            x_next = torch.zeros_like(x)
            count = torch.zeros_like(x)
            eps_noise = torch.randn_like(x)
            if shared:
                dist.broadcast(eps_noise, src=src, group=self.process_group)
            for gi, j0 in enumerate(range(0, len(tiles), kb)):
                if world > 1 and gi % world != rank:
                    continue
                grp = tiles[j0:j0 + kb]
                k = len(grp)
                if k == 1:
                    hi, he, wi, we = grp[0]
                    cj = conds[j0] if use_local_prompt else conds[0]
                    ctl = lq[:, :, hi:he, wi:we]
                    cj["control"] = ctl
                    uc["control"] = ctl
                    cat = dict(static[j0 if use_local_prompt else 0])
                    cat["control"] = torch.cat((ctl, ctl), 0) if not isinstance(self.guider, IdentityGuider) else ctl
                    _x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x[:, :, hi:he, wi:we], cj, uc, gamma,
                                           x_center[:, :, hi:he, wi:we], eps_noise=eps_noise[:, :, hi:he, wi:we],
                                           control_scale=control_scale, use_linear_control_scale=use_linear_control_scale,
                                           control_scale_start=control_scale_start, cond_cat=cat, sigma_f=sf[i],
                                           next_sigma_f=sf[i + 1])
                    outs = [_x]
                else:
                    def stack(t):
                        return torch.cat([t[:, :, a:b_, c:d] for (a, b_, c, d) in grp], 0)
                    ctl = stack(lq)
                    cat = self._static_rep(static[0], k)
                    cat["control"] = torch.cat((ctl, ctl), 0) if not isinstance(self.guider, IdentityGuider) else ctl
                    s_k = s_in.repeat(k)
                    _x = self.sampler_step(s_k * sigmas[i], s_k * sigmas[i + 1], denoiser, stack(x), conds[0], uc, gamma,
                                           stack(x_center), eps_noise=stack(eps_noise), control_scale=control_scale,
                                           use_linear_control_scale=use_linear_control_scale,
                                           control_scale_start=control_scale_start, cond_cat=cat, sigma_f=sf[i],
                                           next_sigma_f=sf[i + 1])
                    outs = list(_x.chunk(k, 0))
                for (hi, he, wi, we), _xt in zip(grp, outs):
                    x_next[:, :, hi:he, wi:we] += _xt * tile_weights
                    count[:, :, hi:he, wi:we] += tile_weights
            if shared:
                dist.all_reduce(x_next, op=dist.ReduceOp.SUM, group=self.process_group)
                count = count_all
            x_next /= count
            x = x_next
        return x

    def _call_fused(self, ctx, x, tiles, tile_weights, lq, static, sf, num_sigmas, x_center, control_scale, use_linear_control_scale,
                    control_scale_start, kb, shared, world, rank, src):
        """The loop of __call__ (sampling.py:624-659) on the fused step: per tile group ONE supir_edm_step_pre_tiles (crop of x and of
        the step's churn noise + the `pre` half), the network call -- announced to the per-image embedding schedule prepared once per
        group size, replayed from one hipGraph per group shape --, supir_edm_step_post on the stacked tiles and ONE supir_tile_blend into
        x_next.  What does not depend on the step is built once per image: the normalising `count` canvas (tile geometry only; the
        reference rebuilds it every step, :629,657), every group's stacked LQ-latent windows (c / uc `control`), its x_center windows and
        its repeated text / vector conditioning.  Tile order, RNG consumption (one randn_like(x) per step) and the per-element
        arithmetic are the generic path's: same results (tests/test_sampler_fused_gpu.py)."""
        import torch.distributed as dist
        from .. import ops
        T = self.tile_size
        twice = not isinstance(self.guider, IdentityGuider)
        groups = [tiles[j0:j0 + kb] for j0 in range(0, len(tiles), kb)]
        mine = [gi for gi in range(len(groups)) if world == 1 or gi % world == rank]
        w64 = tile_weights[0, 0].contiguous()
        count = torch.zeros_like(x)
        for (hi, he, wi, we) in tiles:
            count[:, :, hi:he, wi:we] += tile_weights

        def stack(t, grp):
            return torch.cat([t[:, :, a:b_, c:d] for (a, b_, c, d) in grp], 0)

        gstat = {}
        for gi in mine:
            grp = groups[gi]
            k = len(grp)
            ctl = stack(lq, grp)
            cat = self._static_rep(static, k) if k > 1 else dict(static)
            cat["control"] = torch.cat((ctl, ctl), 0) if twice else ctl
            gstat[gi] = (cat, stack(x_center, grp).float().contiguous())
        sched = False
        if mine and EMB_SCHEDULE and hasattr(ctx[1], "prepare_schedule") and "vector" in static:
            f32 = np.float32
            t_all = [ctx[0].host_scalars(f32(sf[i]) * f32(self._gamma(sf[i], num_sigmas) + 1.0))[0] for i in range(num_sigmas - 1)]
            seen = set()
            for gi in mine:      # one table per group size (all tiles share the timestep); the LQ windows differ per group: no cached hint
                k = len(groups[gi])
                if k not in seen:
                    seen.add(k)
                    ctx[1].prepare_schedule(t_all, gstat[gi][0]["vector"])
            sched = True
        try:
            for i in range(num_sigmas - 1):
                gamma = self._gamma(sf[i], num_sigmas)
                x_next = torch.zeros_like(x)
                eps_noise = torch.randn_like(x)
                if shared:
                    dist.broadcast(eps_noise, src=src, group=self.process_group)
                for gi in mine:
                    cat, xc = gstat[gi]
                    out = self._fused_step(ctx, sf[i], sf[i + 1], x, gamma, xc, eps_noise, control_scale, use_linear_control_scale,
                                           control_scale_start, cat, sched_row=i if sched else None, tiles=(groups[gi], T))
                    ops.tile_blend(out, w64, x_next, groups[gi], T)
                if shared:
                    dist.all_reduce(x_next, op=dist.ReduceOp.SUM, group=self.process_group)
                x_next /= count
                x = x_next
        finally:
            if sched:
                ctx[1].end_schedule()
        return x

    def _static_rep(self, cat, k):
        """[uncond; cond] text / vector conditioning repeated for k stacked tiles: [uc]*k ; [c]*k (cached per k, so the
        network's per-context caches and its hipGraph see the same tensors every step)."""
        cache = self.__dict__.setdefault("_rep_cache", {})
        key = (id(cat), k)
        if key not in cache:
            rep = {}
            for name, v in cat.items():
                if name == "control" or not torch.is_tensor(v):
                    continue
                u, c = v.chunk(2, 0) if not isinstance(self.guider, IdentityGuider) else (v, None)
                rep[name] = torch.cat([u.repeat(k, *([1] * (v.dim() - 1))), c.repeat(k, *([1] * (v.dim() - 1)))], 0).contiguous() \
                    if c is not None else u.repeat(k, *([1] * (v.dim() - 1))).contiguous()
            cache[key] = (cat, rep)   # keep `cat` alive: the key uses its id
        return dict(cache[key][1])


# ----------------------------------------------------------------------------------------------- DPM++ 2M restore sampler
def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """Karras et al. (2022) schedule as in k-diffusion 0.1.1.post1 `get_sigmas_karras` (third-party, not in the reference tree:
    requirements.txt:41): n sigmas from sigma_max to sigma_min on a rho-warped ramp, then 0.  The reference passes
    sigmas[-2].cpu() / sigmas[0].cpu() (sampling.py:490-491): fp32 0-dim tensors, so the roots and the power are fp32 tensor
    arithmetic on the host -- evaluated the same way here (floats are taken as fp32 values).  Checked against the oracle's own
    restatement of the published function (tests/test_host_logic.py::test_karras_schedule_vs_the_oracles_restatement)."""
    ramp = torch.linspace(0, 1, n, dtype=torch.float32)
    smin = torch.as_tensor(sigma_min, dtype=torch.float32).cpu()
    smax = torch.as_tensor(sigma_max, dtype=torch.float32).cpu()
    min_inv_rho = smin ** (1 / rho)
    max_inv_rho = smax ** (1 / rho)
    sig = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sig, sig.new_zeros([1])]).to(device)


class IntervalNoiseSampler:
    """A fresh torch.randn per call: what a Brownian path gives on CONSECUTIVE, non-overlapping sigma intervals (its normalised
    increments there are i.i.d. N(0, 1)) without the path.  The default of rounds 1-4; since round 5 the samplers construct
    brownian.BrownianTreeNoiseSampler (a virtual Brownian tree behind k-diffusion's interface) as the reference does, and this class is
    kept for callers that want the cheaper stream (noise_sampler_cls=IntervalNoiseSampler)."""

    def __init__(self, x, sigma_min=None, sigma_max=None, seed=None):
        self.shape, self.device, self.dtype = x.shape, x.device, x.dtype
        self.gen = None
        if seed is not None:
            self.gen = torch.Generator(device=x.device).manual_seed(seed)

    def __call__(self, sigma, sigma_next):
        return torch.randn(self.shape, device=self.device, dtype=self.dtype, generator=self.gen)


def default_noise_sampler_cls():
    """The class the DPM++ samplers construct their noise source from (sampling.py:494, 687).  The reference imports
    `k_diffusion.sampling.BrownianTreeNoiseSampler` (k-diffusion 0.1.1.post1 over torchsde, requirements.txt:41): when BOTH packages import
    in this process -- a user running with the reference's own requirements installed -- that very class is used, so identical seeds give
    the reference's exact noise stream.  Otherwise (this image: neither is installable) `brownian.BrownianTreeNoiseSampler`, the restatement
    of the published virtual Brownian tree behind the same interface (stream parity with torchsde unpinned: DESIGN section 4).
    SUPIR_BROWNIAN=native forces the restatement."""
    import os
    if os.environ.get("SUPIR_BROWNIAN", "auto") != "native":
        try:
            import torchsde  # noqa: F401  (k-diffusion's class is a thin wrapper over torchsde.BrownianTree)
            from k_diffusion.sampling import BrownianTreeNoiseSampler as theirs
            if callable(theirs):
                return theirs
        except Exception:   # noqa: BLE001 -- absent or broken third-party packages: the restatement serves
            pass
    return BrownianTreeNoiseSampler


def _neg_log(s):
    return s.log().neg()


class RestoreDPMPP2MSampler(BaseDiffusionSampler):
    """sampling.py:422-515 (RestoreDPMPP2MSampler over DPMPP2MSampler :291-362): DPM-Solver++(2M) SDE with Karras sigmas
    between the DDPM schedule's end points; used by options/SUPIR_v0_Juggernautv9_lightning.yaml (8 / 4 steps)."""

    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, restore_cfg=4.0,
                 restore_cfg_s_tmin=0.05, eta=1.0, noise_sampler_cls=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_noise, self.eta = s_noise, eta
        self.noise_sampler_cls = noise_sampler_cls or default_noise_sampler_cls()      # sampling.py:494, 687

    def denoise(self, x, denoiser, sigma, cond, uc, control_scale=1.0, cond_cat=None):
        if cond_cat is None:
            cond_cat = self.guider.prepare_cond(cond, uc)
        twice = not isinstance(self.guider, IdentityGuider)
        xin = torch.cat([x] * 2) if twice else x
        sin = torch.cat([sigma] * 2) if twice else sigma
        return self.guider(denoiser(xin, sin, cond_cat, control_scale), sigma)

    def get_variables(self, sigma, next_sigma, previous_sigma=None):
        t, t_next = _neg_log(sigma), _neg_log(next_sigma)
        h = t_next - t
        if previous_sigma is not None:
            return h, (t - _neg_log(previous_sigma)) / h, t, t_next
        return h, None, t, t_next

    def get_mult(self, h, r, t, t_next, previous_sigma):
        eta_h = self.eta * h
        mult1 = t_next.neg().exp() / t.neg().exp() * (-eta_h).exp()
        mult2 = (-h - eta_h).expm1()
        if previous_sigma is not None:
            return mult1, mult2, 1 + 1 / (2 * r), 1 / (2 * r)
        return mult1, mult2

    def sampler_step(self, old_denoised, previous_sigma, sigma, next_sigma, denoiser, x, cond, uc=None, eps_noise=None,
                     control_scale=1.0, cond_cat=None, last=False):
        denoised = self.denoise(x, denoiser, sigma, cond, uc, control_scale=control_scale, cond_cat=cond_cat)
        h, r, t, t_next = self.get_variables(sigma, next_sigma, previous_sigma)
        eta_h = self.eta * h
        mult = [append_dims(m, x.ndim) for m in self.get_mult(h, r, t, t_next, previous_sigma)]
        x_standard = mult[0] * x - mult[1] * denoised
        if old_denoised is None or last:
            return x_standard, denoised
        denoised_d = mult[2] * denoised - mult[3] * old_denoised
        x = mult[0] * x - mult[1] * denoised_d          # next_sigma > 0 here (the `last` step returned above)
        if self.eta:
            # the reference multiplies [B,4,H,W] by the [B] vector next_sigma (sampling.py:482: only valid for B = 1);
            # append_dims is the same thing for B = 1 and the intended broadcast for B > 1
            x = x + eps_noise * append_dims(next_sigma * (-2 * eta_h).expm1().neg().sqrt(), x.ndim) * self.s_noise
        return x, denoised

    def _karras(self, x, cond, uc, num_steps):
        x, s_in, sigmas, num_sigmas, cond, uc, sf = self.prepare_sampling_loop(x, cond, uc, num_steps)
        # the reference builds the Karras schedule from self.num_steps even when the call names another count (sampling.py:491, 685: the
        # loop length follows the call, the schedule the constructor -- a longer call then indexes past the schedule and raises, a shorter
        # one stops before sigma 0); batchify_sample always calls with num_steps=None (SUPIR_model.py:100, 128), where the two agree
        n = self.num_steps
        smin, smax = sf[-2], sf[0]
        sig_host = get_sigmas_karras(n, smin, smax, device="cpu")
        return x, s_in, sig_host.to(x.device), num_sigmas, cond, uc, [float(v) for v in sig_host], smin, smax

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, control_scale=1.0, **kwargs):
        x, s_in, sigmas, num_sigmas, cond, uc, sf, smin, smax = self._karras(x, cond, uc, num_steps)
        cond_cat = self.guider.prepare_cond(cond, uc)
        noise_sampler = self.noise_sampler_cls(x, smin, smax)
        old = None
        for i in range(num_sigmas - 1):
            eps = noise_sampler(s_in * sigmas[i], s_in * sigmas[i + 1]) if (i > 0 and sf[i + 1] > 1e-14) else None
            x, old = self.sampler_step(old, None if i == 0 else s_in * sigmas[i - 1], s_in * sigmas[i], s_in * sigmas[i + 1],
                                       denoiser, x, cond, uc=uc, eps_noise=eps, control_scale=control_scale,
                                       cond_cat=cond_cat, last=sf[i + 1] < 1e-14)
        return x


class TiledRestoreDPMPP2MSampler(RestoreDPMPP2MSampler):
    """sampling.py:663-730: tile loop + Gaussian blend around the DPM++ step (x and old_denoised are both blended)."""

    def __init__(self, tile_size=128, tile_stride=64, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tile_size, self.tile_stride = tile_size, tile_stride

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, control_scale=1.0, **kwargs):
        use_local_prompt = isinstance(cond, list)
        b, _, h, w = x.shape
        tiles = _sliding_windows(h, w, self.tile_size, self.tile_stride)
        tile_weights = gaussian_weights(self.tile_size, self.tile_size, 1, device=x.device).repeat(b, 1, 1, 1)
        lq = cond[0]["control"] if use_local_prompt else cond["control"]
        x, s_in, sigmas, num_sigmas, cond, uc, sf, smin, smax = self._karras(x, cond, uc, num_steps)
        uc = dict(uc)
        conds = [dict(cj) for cj in cond] if use_local_prompt else [dict(cond)]
        static = [self.guider.prepare_cond({k: v for k, v in cj.items() if k != "control"},
                                           {k: v for k, v in uc.items() if k != "control"}) for cj in conds]
        noise_sampler = self.noise_sampler_cls(x, smin, smax)
        old = None
        for i in range(num_sigmas - 1):
            eps = noise_sampler(s_in * sigmas[i], s_in * sigmas[i + 1]) if (i > 0 and sf[i + 1] > 1e-14) else torch.zeros_like(x)
            x_next, old_next, count = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
            for j, (hi, he, wi, we) in enumerate(tiles):
                cj = conds[j] if use_local_prompt else conds[0]
                ctl = lq[:, :, hi:he, wi:we]
                cj["control"] = ctl
                uc["control"] = ctl
                cat = dict(static[j if use_local_prompt else 0])
                cat["control"] = torch.cat((ctl, ctl), 0) if not isinstance(self.guider, IdentityGuider) else ctl
                _x, _old = self.sampler_step(None if old is None else old[:, :, hi:he, wi:we],
                                             None if i == 0 else s_in * sigmas[i - 1], s_in * sigmas[i], s_in * sigmas[i + 1],
                                             denoiser, x[:, :, hi:he, wi:we], cj, uc=uc, eps_noise=eps[:, :, hi:he, wi:we],
                                             control_scale=control_scale, cond_cat=cat, last=sf[i + 1] < 1e-14)
                x_next[:, :, hi:he, wi:we] += _x * tile_weights
                old_next[:, :, hi:he, wi:we] += _old * tile_weights
                count[:, :, hi:he, wi:we] += tile_weights
            old = old_next / count
            x = x_next / count
        return x
