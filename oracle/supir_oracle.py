"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (supir_amd/), only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

A plain-PyTorch fp32, NCHW, functional restatement of SUPIR's restoration-guided EDM sampling path, written against a
flat {reference state-dict key: tensor} dict.  It is the checker for the HIP path; it is pinned against the real
reference: tests/golden/*.pt were produced by importing /root/reference (oracle/gen_golden.py, run in the build
container where the reference is mounted) and tests/test_oracle_golden.py checks this file against them.

Every function cites the reference code it restates (paths relative to the reference root).

Third-party arithmetic on the path that is NOT in the reference tree and not installable in the build container -- "parity unpinned"
against the packages themselves, restated from what they publish:
  * k-diffusion 0.1.1.post1 (requirements.txt:41) `get_sigmas_karras`: restated below (kdiff_get_sigmas_karras), the DPM++ goldens
    are generated with it, the product's function is held bitwise to it;
  * k-diffusion `BrownianTreeNoiseSampler` -> torchsde `BrownianTree` (sampling.py:494, 687): no oracle counterpart -- a noise source has
    no arithmetic to compare, only a stream; the product restates the published virtual-Brownian-tree algorithm
    (supir_amd/modules/brownian.py) and tests/test_brownian.py holds its defining properties; the DPM++ solver is pinned with scripted noise;
  * open_clip 2.17.1 (bigG text tower): oracle/cond_oracle.py, held against transformers.CLIPTextModelWithProjection
    (tests/test_conditioner.py).
The structure (depths, which blocks have transformers, channel counts) is read off the key names / tensor shapes,
so the same code runs the full SDXL-sized model and the reduced-depth models used for CPU-sized tests.
"""
import math

import torch
import torch.nn.functional as F

SIGMA_MAX = 14.6146  # sgm/modules/diffusionmodules/sampling.py:541, guiders.py:48


# ----------------------------------------------------------------------------------------------- primitives
def timestep_embedding(t, dim, max_period=10000):
    """cos || sin, freqs = exp(-ln(max_period) * i / half)  (sgm/modules/diffusionmodules/util.py:206-230)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mlp_embed(sd, p, x):
    """Linear -> SiLU -> Linear (time_embed / label_emb.0; openaimodel.py:665-696)."""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


# ----------------------------------------------------------------------------------------------- UNet blocks
def res_block(sd, p, x, emb):
    """ResBlock._forward without up/down, no scale-shift (openaimodel.py:330-356).
    in_layers = [GN(eps 1e-5), SiLU, conv3x3]; emb_layers = [SiLU, Linear]; out_layers = [GN, SiLU, Dropout, conv3x3]."""
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)))
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward (sgm/modules/attention.py:222-285): to_q/k/v no bias, 'b n (h d) -> b h n d',
    softmax(q k^T / sqrt(d)) v, to_out.0 with bias."""
    context = x if context is None else context
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", context), _lin(sd, p + ".to_v", context)
    b, n, c = q.shape
    d = c // heads
    q, k, v = (t.reshape(b, -1, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(sd, p + ".to_out.0", o)


def geglu_ff(sd, p, x):
    """FeedForward(glu=True): GEGLU (value = first half, gate = second half, erf GELU) then Linear
    (attention.py:84-110)."""
    v, g = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", v * F.gelu(g))


def basic_transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward (attention.py:465-486)."""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = geglu_ff(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd, p, x, context):
    """SpatialTransformer.forward with use_linear=True (attention.py:614-635): GN(eps 1e-6) -> 'b c h w -> b (h w) c'
    -> proj_in -> blocks -> proj_out -> back -> + x_in.  heads = C / 64 (num_head_channels 64)."""
    b, c, h, w = x.shape
    heads = c // 64
    x_in = x
    t = _gn(sd, p + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = _lin(sd, p + ".proj_in", t)
    i = 0
    while f"{p}.transformer_blocks.{i}.attn1.to_q.weight" in sd:
        t = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", t, context, heads)
        i += 1
    t = _lin(sd, p + ".proj_out", t)
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x_in


def _is_res(sd, p):
    return (p + ".in_layers.0.weight") in sd


def _is_st(sd, p):
    return (p + ".proj_in.weight") in sd


def timestep_embed_sequential(sd, p, x, emb, context):
    """TimestepEmbedSequential.forward (openaimodel.py:87-105) over children p.0, p.1, ...:
    ResBlock(x, emb) / SpatialTransformer(x, context) / Downsample.op (conv s2 p1) / Upsample (nearest 2x + conv) /
    plain conv."""
    i = 0
    while True:
        q = f"{p}.{i}"
        if _is_res(sd, q):
            x = res_block(sd, q, x, emb)
        elif _is_st(sd, q):
            x = spatial_transformer(sd, q, x, context)
        elif (q + ".op.weight") in sd:  # Downsample (openaimodel.py:196-210)
            x = _conv(sd, q + ".op", x, stride=2, padding=1)
        elif (q + ".conv.weight") in sd:  # Upsample (openaimodel.py:131-151); the fp32 round trip is a no-op in fp32
            x = _conv(sd, q + ".conv", F.interpolate(x, scale_factor=2, mode="nearest"))
        elif (q + ".weight") in sd:  # bare conv (input_blocks.0.0, input_hint_block.0)
            x = _conv(sd, q, x)
        else:
            break
        i += 1
    return x


def _embed(sd, p, t, y):
    """emb = time_embed(timestep_embedding(t, 320)) + label_emb(y)  (SUPIR_v0.py:515-522, 618-623)."""
    mc = sd[p + "time_embed.0.weight"].shape[1]
    emb = _mlp_embed(sd, p + "time_embed", timestep_embedding(t, mc))
    return emb + _mlp_embed(sd, p + "label_emb.0", y)


# ----------------------------------------------------------------------------------------------- SUPIR control / adapters
def glv_control(sd, x, timesteps, xt, context, y, p="control_model.", taps=None):
    """GLVControl.forward (SUPIR/modules/SUPIR_v0.py:499-540): SDXL encoder + middle on xt, the LQ latent enters through
    input_hint_block (added after input block 0); returns the 10 feature maps."""
    emb = _embed(sd, p, timesteps, y)
    guided_hint = timestep_embed_sequential(sd, p + "input_hint_block", x, emb, context)
    hs = []
    h = xt
    i = 0
    while f"{p}input_blocks.{i}.0.weight" in sd or _is_res(sd, f"{p}input_blocks.{i}.0") or \
            f"{p}input_blocks.{i}.0.op.weight" in sd:
        h = timestep_embed_sequential(sd, f"{p}input_blocks.{i}", h, emb, context)
        if guided_hint is not None:
            h = h + guided_hint
            guided_hint = None
        hs.append(h)
        i += 1
    h = timestep_embed_sequential(sd, p + "middle_block", h, emb, context)
    hs.append(h)
    return hs


def zero_sft(sd, p, c, h, h_ori=None, control_scale=1.0):
    """ZeroSFT.forward (SUPIR_v0.py:91-113); pre_concat == (h_ori is not None) for every instance LightGLVUNet builds."""
    h_raw = torch.cat([h_ori, h], dim=1) if h_ori is not None else h
    h = h + _conv(sd, p + ".zero_conv", c, padding=0)
    if h_ori is not None:
        h = torch.cat([h_ori, h], dim=1)
    actv = F.silu(_conv(sd, p + ".mlp_shared.0", c))
    gamma = _conv(sd, p + ".zero_mul", actv)
    beta = _conv(sd, p + ".zero_add", actv)
    h = _gn(sd, p + ".param_free_norm", h, 1e-5) * (gamma + 1) + beta
    return h * control_scale + h_raw * (1 - control_scale)


def zero_cross_attn(sd, p, context, x, control_scale=1.0):
    """ZeroCrossAttn.forward (SUPIR_v0.py:138-152): q from GN(x), k/v from GN(control feature), heads = C/64."""
    b, c, h, w = x.shape
    xn = _gn(sd, p + ".norm1", x, 1e-5).permute(0, 2, 3, 1).reshape(b, h * w, c)
    cn = _gn(sd, p + ".norm2", context, 1e-5)
    cn = cn.permute(0, 2, 3, 1).reshape(b, h * w, cn.shape[1])
    o = cross_attention(sd, p + ".attn", xn, cn, c // 64)
    return x + o.reshape(b, h, w, c).permute(0, 3, 1, 2) * control_scale


def light_glv_unet(sd, x, timesteps, context, y, control, control_scale=1.0, p="diffusion_model.", taps=None):
    """LightGLVUNet.forward (SUPIR_v0.py:600-666): encoder, middle, then for every output block the skip concat is
    replaced by project_modules[adapter_idx] (ZeroSFT), with a ZeroCrossAttn before the Upsample of 3-child blocks.
    taps (a dict, tests only): filled with every module boundary tensor -- 'enc{i}' (input block outputs), 'mid', 'adapter{k}'
    (project_modules[k] outputs), 'res{i}' / 'st{i}' / 'out{i}' (children / result of output block i) -- for teacher-forced
    per-block checks of the product modules."""
    tp = taps if taps is not None else {}
    emb = _embed(sd, p, timesteps, y)
    hs = []
    h = x
    i = 0
    while f"{p}input_blocks.{i}.0.weight" in sd or _is_res(sd, f"{p}input_blocks.{i}.0") or \
            f"{p}input_blocks.{i}.0.op.weight" in sd:
        h = timestep_embed_sequential(sd, f"{p}input_blocks.{i}", h, emb, context)
        hs.append(h)
        tp[f"enc{i}"] = h
        i += 1
    n_proj = 0
    while f"{p}project_modules.{n_proj}.zero_conv.weight" in sd or f"{p}project_modules.{n_proj}.norm1.weight" in sd:
        n_proj += 1
    adapter_idx, control_idx = n_proj - 1, len(control) - 1
    h = timestep_embed_sequential(sd, p + "middle_block", h, emb, context)
    tp["mid"] = h
    h = zero_sft(sd, f"{p}project_modules.{adapter_idx}", control[control_idx], h, control_scale=control_scale)
    tp[f"adapter{adapter_idx}"] = h
    adapter_idx -= 1
    control_idx -= 1
    i = 0
    while _is_res(sd, f"{p}output_blocks.{i}.0"):
        q = f"{p}output_blocks.{i}"
        _h = hs.pop()
        h = zero_sft(sd, f"{p}project_modules.{adapter_idx}", control[control_idx], _h, h, control_scale=control_scale)
        tp[f"adapter{adapter_idx}"] = h
        adapter_idx -= 1
        if (q + ".2.conv.weight") in sd:  # [Res, ST, Upsample]
            h = res_block(sd, q + ".0", h, emb)
            tp[f"res{i}"] = h
            h = spatial_transformer(sd, q + ".1", h, context)
            tp[f"st{i}"] = h
            h = zero_cross_attn(sd, f"{p}project_modules.{adapter_idx}", control[control_idx], h, control_scale)
            tp[f"adapter{adapter_idx}"] = h
            adapter_idx -= 1
            h = _conv(sd, q + ".2.conv", F.interpolate(h, scale_factor=2, mode="nearest"))
        else:
            h = timestep_embed_sequential(sd, q, h, emb, context)
        tp[f"out{i}"] = h
        control_idx -= 1
        i += 1
    # self.out = [GN, SiLU, conv3x3]  (openaimodel.py:947-953)
    return _conv(sd, p + "out.2", F.silu(_gn(sd, p + "out.0", h, 1e-5)))


def control_wrapper(sd, x, t, c, control_scale=1.0, p="model."):
    """ControlWrapper.forward (sgm/modules/diffusionmodules/wrappers.py:84-102), fp32 (autocast is a no-op on CPU)."""
    control = glv_control(sd, c["control"], t, x, c["crossattn"], c["vector"], p=p + "control_model.")
    out = light_glv_unet(sd, x, t, c["crossattn"], c["vector"], control, control_scale, p=p + "diffusion_model.")
    return out.float()


# ----------------------------------------------------------------------------------------------- denoiser / sampler
def kdiff_append_zero(x):
    """k-diffusion 0.1.1.post1 `append_zero` (k_diffusion/sampling.py; third-party, requirements.txt:41 -- not in the reference tree)."""
    return torch.cat([x, x.new_zeros([1])])


def kdiff_get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """k-diffusion 0.1.1.post1 `get_sigmas_karras` (k_diffusion/sampling.py, "Constructs the noise schedule of Karras et al.
    (2022)"), restated from the PUBLISHED package source -- third-party (requirements.txt:41), absent from /root/reference and not
    installable here.  Called by the reference at sgm/modules/diffusionmodules/sampling.py:491-492 and :684-685 with
    sigma_min = sigmas[-2].cpu(), sigma_max = sigmas[0].cpu() (0-dim fp32 tensors) and device = x.device: the ramp is a default-dtype
    (fp32) linspace on the CPU, the rho-th roots and the rho-th power are fp32 TENSOR pows, and the zero is appended before the move.
    Independent of supir_amd.modules.sampling.get_sigmas_karras, which tests/test_host_logic.py checks against THIS function.
    Still third-party: "parity unpinned" against an installed k-diffusion, pinned against the published FORMULA only.  In particular the
    published releases may build the ramp with `torch.linspace(0, 1, n, device=device)`, i.e. evaluate the linspace and both pows on
    x.device (the GPU) where this restatement evaluates them on the host: the two differ by at most an ulp of fp32 per sigma, and which of
    them a given k-diffusion release does cannot be checked here -- "held bitwise" in tests means bitwise to THIS restatement."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return kdiff_append_zero(sigmas).to(device)


def ddpm_sigmas(n=1000, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000, device="cpu"):
    """LegacyDDPMDiscretization (sgm/modules/diffusionmodules/discretizer.py:42-69) incl. make_beta_schedule('linear')
    = linspace(sqrt(start), sqrt(end), n)**2 in float64 (util.py:25-33). Returned in INCREASING-t order flipped, i.e.
    sigmas[0] is the largest (what `Discretization.__call__(flip=False)` yields before append_zero)."""
    import numpy as np
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas, axis=0)
    if n < num_timesteps:
        ts = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        ac = ac[ts]
    elif n != num_timesteps:
        raise ValueError
    sig = torch.tensor((1 - ac) / ac, dtype=torch.float32, device=device) ** 0.5
    return torch.flip(sig, (0,))


def denoiser_table(device="cpu"):
    """DiscreteDenoiser.sigmas buffer: 1000 sigmas, flip=True -> increasing (denoiser.py:31-47)."""
    return torch.flip(ddpm_sigmas(1000, device=device), (0,))


def discrete_denoiser_with_control(network, table, x, sigma, cond, control_scale):
    """DiscreteDenoiserWithControl.__call__ (denoiser.py:66-73) with EpsScaling (denoiser_scaling.py:16-22):
    sigma snapped to the table, c_skip = 1, c_out = -sigma, c_in = 1/sqrt(sigma^2+1), c_noise = table index (int64)."""
    idx = (sigma[None, :] - table[:, None]).abs().argmin(dim=0)
    sig = table[idx]
    s4 = sig.view(-1, 1, 1, 1)
    c_in = 1 / (s4 ** 2 + 1.0) ** 0.5
    c_noise = (sig[None, :] - table[:, None]).abs().argmin(dim=0)
    return network(x * c_in, c_noise, cond, control_scale) * (-s4) + x


def linear_cfg_scale(scale, scale_min, sigma):
    """LinearCFG.scale_schedule (guiders.py:45-49)."""
    return (scale - scale_min) * sigma / SIGMA_MAX + scale_min


def guided_denoise(denoise_fn, x, sigma_hat, cond, uc, scale, scale_min, control_scale):
    """RestoreEDMSampler.denoise + LinearCFG.prepare_inputs/__call__ (sampling.py:543-546, guiders.py:59-74):
    batch = [uncond; cond]; the guider sees the UN-quantised sigma_hat."""
    c_out = {}
    for k in cond:
        c_out[k] = torch.cat((uc[k], cond[k]), 0)
    den = denoise_fn(torch.cat([x] * 2), torch.cat([sigma_hat] * 2), c_out, control_scale)
    x_u, x_c = den.chunk(2)
    s = linear_cfg_scale(scale, scale_min, sigma_hat)
    return x_u + s.view(-1, 1, 1, 1) * (x_c - x_u)


def restore_edm_step(denoise_fn, x, sigma, next_sigma, gamma, cond, uc, x_center, eps, *, s_noise, restore_cfg,
                     restore_cfg_s_tmin=0.05, scale=1.0, scale_min=4.0, control_scale=1.0):
    """RestoreEDMSampler.sampler_step (sampling.py:548-570). `eps` is the churn noise tensor (injected so CPU-oracle
    and GPU runs see identical noise; the reference draws it with torch.randn_like, :555)."""
    sigma_hat = sigma * (gamma + 1.0)
    if gamma > 0:
        x = x + eps * s_noise * (sigma_hat ** 2 - sigma ** 2).view(-1, 1, 1, 1) ** 0.5
    den = guided_denoise(denoise_fn, x, sigma_hat, cond, uc, scale, scale_min, control_scale)
    if (next_sigma[0] > restore_cfg_s_tmin) and (restore_cfg > 0):
        d_center = den - x_center
        den = den - d_center * ((sigma.view(-1, 1, 1, 1) / SIGMA_MAX) ** restore_cfg)
    d = (x - den) / sigma_hat.view(-1, 1, 1, 1)
    return x + d * (next_sigma - sigma_hat).view(-1, 1, 1, 1)


def restore_edm_sample(denoise_fn, x, cond, uc, x_center, noises, *, num_steps, s_churn, s_noise, restore_cfg,
                       scale=1.0, scale_min=4.0, control_scale=1.0, s_tmin=0.0, s_tmax=float("inf"),
                       use_linear_control_scale=False, control_scale_start=0.0):
    """RestoreEDMSampler.__call__ + prepare_sampling_loop (sampling.py:45-56, 572-597). noises[i] = churn noise of step i.
    use_linear_control_scale (sampling.py:557-559): the control scale of a step is interpolated on sigma / sigma_max between
    `control_scale` (sigma -> 0) and `control_scale_start` (sigma = sigma_max)."""
    sigmas = torch.cat([ddpm_sigmas(num_steps, device=x.device), x.new_zeros([1])])
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    n = len(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(n - 1):
        gamma = min(s_churn / (n - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
        cs = control_scale
        if use_linear_control_scale:
            cs = (float(sigmas[i]) / SIGMA_MAX) * (control_scale_start - control_scale) + control_scale
        x = restore_edm_step(denoise_fn, x, s_in * sigmas[i], s_in * sigmas[i + 1], gamma, cond, uc, x_center,
                             noises[i] if noises is not None else None, s_noise=s_noise, restore_cfg=restore_cfg,
                             scale=scale, scale_min=scale_min, control_scale=cs)
    return x


def gaussian_weights(tile_width, tile_height):
    """gaussian_weights (sampling.py:733-750): float64, var 0.01, x midpoint (w-1)/2, y midpoint h/2 (asymmetric on
    purpose: reproduces the reference). Returns [1, 4, h, w] float64."""
    import numpy as np
    from numpy import exp, pi, sqrt
    var = 0.01
    mid = (tile_width - 1) / 2
    xp = [exp(-(x - mid) * (x - mid) / (tile_width * tile_width) / (2 * var)) / sqrt(2 * pi * var)
          for x in range(tile_width)]
    mid = tile_height / 2
    yp = [exp(-(y - mid) * (y - mid) / (tile_height * tile_height) / (2 * var)) / sqrt(2 * pi * var)
          for y in range(tile_height)]
    return torch.tile(torch.tensor(np.outer(yp, xp)), (1, 4, 1, 1))


def sliding_windows(h, w, tile_size, tile_stride):
    """_sliding_windows (sampling.py:753-766)."""
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


def tiled_restore_edm_sample(denoise_fn, x, cond, uc, x_center, noises, *, tile_size, tile_stride, num_steps, s_churn,
                             s_noise, restore_cfg, scale=1.0, scale_min=4.0, control_scale=1.0):
    """TiledRestoreEDMSampler.__call__ (sampling.py:607-660), global prompt (cond is a dict). noises[i] is the full-size
    per-step eps_noise (:631), sliced per tile."""
    b, _, h, w = x.shape
    tiles = sliding_windows(h, w, tile_size, tile_stride)
    tw = gaussian_weights(tile_size, tile_size).to(x.device).repeat(b, 1, 1, 1)
    lq = cond["control"]
    sigmas = torch.cat([ddpm_sigmas(num_steps, device=x.device), x.new_zeros([1])])
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    n = len(sigmas)
    s_in = x.new_ones([b])
    cond, uc = dict(cond), dict(uc)
    for i in range(n - 1):
        gamma = min(s_churn / (n - 1), 2 ** 0.5 - 1)
        x_next = torch.zeros_like(x)
        count = torch.zeros_like(x)
        for (hi, he, wi, we) in tiles:
            cond["control"] = lq[:, :, hi:he, wi:we]
            uc["control"] = lq[:, :, hi:he, wi:we]
            _x = restore_edm_step(denoise_fn, x[:, :, hi:he, wi:we], s_in * sigmas[i], s_in * sigmas[i + 1], gamma, cond,
                                  uc, x_center[:, :, hi:he, wi:we], noises[i][:, :, hi:he, wi:we], s_noise=s_noise,
                                  restore_cfg=restore_cfg, scale=scale, scale_min=scale_min, control_scale=control_scale)
            # fp32 += fp32 * fp64 -> computed in fp64, stored back into the fp32 buffer (reference dtype behaviour)
            x_next[:, :, hi:he, wi:we] += _x * tw
            count[:, :, hi:he, wi:we] += tw
        x = x_next / count
    return x


# ----------------------------------------------------------------------------------------------- VAE
def vae_resnet_block(sd, p, x):
    """ResnetBlock.forward with temb=None (sgm/modules/diffusionmodules/model.py:128-148); GN eps 1e-6, swish."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, 1e-6)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def vae_attn_block(sd, p, x):
    """AttnBlock / MemoryEfficientAttnBlock (model.py:158-262): single head over H*W tokens, head dim = C."""
    h_ = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (_conv(sd, f"{p}.{n}", h_, padding=0) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    q, k, v = (t.reshape(b, c, h * w).permute(0, 2, 1)[:, None] for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)[:, 0].permute(0, 2, 1).reshape(b, c, h, w)
    return x + _conv(sd, p + ".proj_out", o, padding=0)


def vae_encoder(sd, x, p="encoder."):
    """Encoder.forward (model.py:571-596); Downsample = F.pad(0,1,0,1) + conv s2 p0 (:81-86)."""
    h = _conv(sd, p + "conv_in", x)
    lvl = 0
    while f"{p}down.{lvl}.block.0.norm1.weight" in sd:
        blk = 0
        while f"{p}down.{lvl}.block.{blk}.norm1.weight" in sd:
            h = vae_resnet_block(sd, f"{p}down.{lvl}.block.{blk}", h)
            blk += 1
        if f"{p}down.{lvl}.downsample.conv.weight" in sd:
            h = _conv(sd, f"{p}down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
        lvl += 1
    h = vae_resnet_block(sd, p + "mid.block_1", h)
    h = vae_attn_block(sd, p + "mid.attn_1", h)
    h = vae_resnet_block(sd, p + "mid.block_2", h)
    return _conv(sd, p + "conv_out", F.silu(_gn(sd, p + "norm_out", h, 1e-6)))


def vae_decoder(sd, z, p="decoder."):
    """Decoder.forward (model.py:710-743); Upsample = nearest 2x + conv (:64-68)."""
    h = _conv(sd, p + "conv_in", z)
    h = vae_resnet_block(sd, p + "mid.block_1", h)
    h = vae_attn_block(sd, p + "mid.attn_1", h)
    h = vae_resnet_block(sd, p + "mid.block_2", h)
    nlev = 0
    while f"{p}up.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lvl in reversed(range(nlev)):
        blk = 0
        while f"{p}up.{lvl}.block.{blk}.norm1.weight" in sd:
            h = vae_resnet_block(sd, f"{p}up.{lvl}.block.{blk}", h)
            blk += 1
        if f"{p}up.{lvl}.upsample.conv.weight" in sd:
            h = _conv(sd, f"{p}up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, p + "conv_out", F.silu(_gn(sd, p + "norm_out", h, 1e-6)))


def vae_moments(sd, x, p="first_stage_model.", encoder="encoder", tile=None):
    """AutoencoderKL.encode up to the moments (sgm/models/autoencoder.py:304-311): encoder -> quant_conv 1x1.
    tile: encoder tile size in pixels when the encoder runs under VAEHook (SUPIRModel.init_tile_vae, SUPIR_model.py:138-150)."""
    h = vae_encoder(sd, x, p + encoder + ".") if tile is None else vae_tiled_forward(sd, x, p + encoder + ".", tile, False)
    return _conv(sd, p + "quant_conv", h, padding=0)


def gaussian_mode_sample(moments, noise=None):
    """DiagonalGaussianDistribution (sgm/modules/distributions/distributions.py:24-41,71-72): mean / mean + std * noise,
    logvar clamped to [-30, 20]."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def vae_decode(sd, z, p="first_stage_model.", tile=None):
    """AutoencoderKL.decode (autoencoder.py:313-316): post_quant_conv 1x1 -> decoder (tile: decoder tile size in latent pixels under
    VAEHook, as vae_moments)."""
    h = _conv(sd, p + "post_quant_conv", z, padding=0)
    return vae_decoder(sd, h, p + "decoder.") if tile is None else vae_tiled_forward(sd, h, p + "decoder.", tile, True)


SCALE_FACTOR = 0.13025  # options/SUPIR_v0.yaml:6


def encode_first_stage_with_denoise(sd, x, tile=None):
    """SUPIRModel.encode_first_stage_with_denoise(use_sample=False) (SUPIR/models/SUPIR_model.py:49-62)."""
    return SCALE_FACTOR * gaussian_mode_sample(vae_moments(sd, x, encoder="denoise_encoder", tile=tile))


def encode_first_stage(sd, x, noise, tile=None):
    """SUPIRModel.encode_first_stage (SUPIR_model.py:42-46): posterior.sample() with the injected noise."""
    return SCALE_FACTOR * gaussian_mode_sample(vae_moments(sd, x, tile=tile), noise)


def decode_first_stage(sd, z, tile=None):
    """SUPIRModel.decode_first_stage (SUPIR_model.py:65-69)."""
    return vae_decode(sd, z / SCALE_FACTOR, tile=tile).float()


def batchify_sample(sd, x, c, uc, noises, *, num_steps, s_churn=5, s_noise=1.01, restoration_scale=-1.0,
                    cfg_scale=4.0, cfg_scale_start=1.0, control_scale=1.0, table=None, tile_vae=None):
    """SUPIRModel.batchify_sample (SUPIR_model.py:80-136) with use_linear_CFG=True, color_fix 'None', conditioner
    bypassed (c / uc given: crossattn, vector).  noises = {'posterior': [N,4,h,w], 'init': [N,4,h,w],
    'steps': [num_steps x [N,4,h,w]]} replaces the three RNG draws (q1 in SURVEY 3.7).
    tile_vae = (encoder tile px, decoder tile latent px): the three VAE nets under VAEHook (test.py --use_tile_vae, test.py:65-66)."""
    table = denoiser_table(x.device) if table is None else table
    te, td = (None, None) if tile_vae is None else tile_vae
    _z = encode_first_stage_with_denoise(sd, x, tile=te)
    x_stage1 = decode_first_stage(sd, _z, tile=td)
    z_stage1 = encode_first_stage(sd, x_stage1, noises["posterior"], tile=te)
    c = dict(c, control=_z)
    uc = dict(uc, control=_z)

    def network(xin, t, cond, cs):
        return control_wrapper(sd, xin, t, cond, cs)

    def denoise_fn(xin, sigma, cond, cs):
        return discrete_denoiser_with_control(network, table, xin, sigma, cond, cs)

    samples = restore_edm_sample(denoise_fn, noises["init"].clone(), c, uc, z_stage1, noises["steps"], num_steps=num_steps,
                                 s_churn=s_churn, s_noise=s_noise, restore_cfg=restoration_scale, scale=cfg_scale_start,
                                 scale_min=cfg_scale, control_scale=control_scale)
    return decode_first_stage(sd, samples, tile=td), dict(z=_z, x_stage1=x_stage1, z_stage1=z_stage1, samples=samples)


def wavelet_reconstruction(content, style):
    """wavelet_reconstruction (SUPIR/utils/colorfix.py:73-119): 5-level dilated 3x3 blur; high freq of content + low
    freq of style."""
    def blur(img, radius):
        k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]],
                         dtype=img.dtype, device=img.device)[None, None].repeat(3, 1, 1, 1)
        img = F.pad(img, (radius,) * 4, mode="replicate")
        return F.conv2d(img, k, groups=3, dilation=radius)

    def decomp(img, levels=5):
        high = torch.zeros_like(img)
        for i in range(levels):
            low = blur(img, 2 ** i)
            high = high + (img - low)
            img = low
        return high, low

    ch, _ = decomp(content)
    _, sl = decomp(style)
    return ch + sl


# ----------------------------------------------------------------------------------------------- tiled VAE
def _best_tile(lowerbound, upperbound):
    """VAEHook.get_best_tile_size (SUPIR/utils/tilevae.py:702-715)."""
    divider = 32
    while divider >= 2:
        rem = lowerbound % divider
        if rem == 0:
            return lowerbound
        cand = lowerbound - rem + divider
        if cand <= upperbound:
            return cand
        divider //= 2
    return lowerbound


def vae_split_tiles(h, w, tile_size, pad, is_decoder):
    """VAEHook.split_tiles (tilevae.py:717-774); bbox = [x1, x2, y1, y2]."""
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    rth = _best_tile(math.ceil((h - 2 * pad) / nh), tile_size)
    rtw = _best_tile(math.ceil((w - 2 * pad) / nw), tile_size)
    in_b, out_b = [], []
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * rtw, min(pad + (j + 1) * rtw, w), pad + i * rth, min(pad + (i + 1) * rth, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            out_b.append([x * 8 if is_decoder else x // 8 for x in ob])
            in_b.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return in_b, out_b


def _pooled_gn(sd, p, tiles, silu):
    """GroupNormParam.add_tile / summary + custom_group_norm (tilevae.py:599-648, 524-553): pixel-weighted mean of the
    per-tile biased variances and means per (batch, group); eps 1e-6."""
    vs, ms, px = [], [], []
    for t in tiles:
        b, c = t.shape[:2]
        r = t.reshape(1, b * 32, c // 32, *t.shape[2:])
        v, m = torch.var_mean(r, dim=[0, 2, 3, 4], unbiased=False)
        vs.append(v)
        ms.append(m)
        px.append(t.shape[2] * t.shape[3])
    w = torch.tensor(px, dtype=torch.float32, device=tiles[0].device) / max(px)
    w = (w / w.sum()).unsqueeze(1)
    var = (torch.vstack(vs) * w).sum(0)
    mean = (torch.vstack(ms) * w).sum(0)
    out = []
    for t in tiles:
        b, c = t.shape[:2]
        r = t.reshape(1, b * 32, c // 32, *t.shape[2:])
        o = F.batch_norm(r, mean, var, training=False, momentum=0, eps=1e-6).reshape(t.shape)
        o = o * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)
        out.append(F.silu(o) if silu else o)
    return out


def _tiled_resblock(sd, p, tiles):
    res = [_conv(sd, p + ".nin_shortcut", t, padding=0) for t in tiles] if (p + ".nin_shortcut.weight") in sd else tiles
    h = [_conv(sd, p + ".conv1", t) for t in _pooled_gn(sd, p + ".norm1", tiles, True)]
    h = [_conv(sd, p + ".conv2", t) for t in _pooled_gn(sd, p + ".norm2", h, True)]
    return [a + b for a, b in zip(h, res)]


def _tiled_attn(sd, p, tiles):
    out = []
    for t, n in zip(tiles, _pooled_gn(sd, p + ".norm", tiles, False)):
        q, k, v = (_conv(sd, f"{p}.{m}", n, padding=0) for m in ("q", "k", "v"))
        b, c, h, w = q.shape
        q, k, v = (x.reshape(b, c, h * w).permute(0, 2, 1)[:, None] for x in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v)[:, 0].permute(0, 2, 1).reshape(b, c, h, w)
        out.append(t + _conv(sd, p + ".proj_out", o, padding=0))
    return out


def vae_tiled_forward(sd, z, p, tile_size, is_decoder):
    """VAEHook.__call__ / vae_tile_forward (tilevae.py:688-700, 821-970) with fast modes off, executed layer-major."""
    pad = 11 if is_decoder else 32
    H, W = z.shape[2], z.shape[3]
    if max(H, W) <= pad * 2 + tile_size:
        return vae_decoder(sd, z, p) if is_decoder else vae_encoder(sd, z, p)
    in_b, out_b = vae_split_tiles(H, W, tile_size, pad, is_decoder)
    tiles = [_conv(sd, p + "conv_in", z[:, :, b[2]:b[3], b[0]:b[1]]) for b in in_b]
    nlev = 0
    key = "up" if is_decoder else "down"
    while f"{p}{key}.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1

    def mid(ts):
        ts = _tiled_resblock(sd, p + "mid.block_1", ts)
        ts = _tiled_attn(sd, p + "mid.attn_1", ts)
        return _tiled_resblock(sd, p + "mid.block_2", ts)

    if is_decoder:
        tiles = mid(tiles)
        for lvl in reversed(range(nlev)):
            blk = 0
            while f"{p}up.{lvl}.block.{blk}.norm1.weight" in sd:
                tiles = _tiled_resblock(sd, f"{p}up.{lvl}.block.{blk}", tiles)
                blk += 1
            if f"{p}up.{lvl}.upsample.conv.weight" in sd:
                tiles = [_conv(sd, f"{p}up.{lvl}.upsample.conv", F.interpolate(t, scale_factor=2.0, mode="nearest")) for t in tiles]
    else:
        for lvl in range(nlev):
            blk = 0
            while f"{p}down.{lvl}.block.{blk}.norm1.weight" in sd:
                tiles = _tiled_resblock(sd, f"{p}down.{lvl}.block.{blk}", tiles)
                blk += 1
            if f"{p}down.{lvl}.downsample.conv.weight" in sd:
                tiles = [_conv(sd, f"{p}down.{lvl}.downsample.conv", F.pad(t, (0, 1, 0, 1)), stride=2, padding=0) for t in tiles]
        tiles = mid(tiles)
    outs = [_conv(sd, p + "conv_out", t) for t in _pooled_gn(sd, p + "norm_out", tiles, True)]
    oh, ow = (H * 8, W * 8) if is_decoder else (H // 8, W // 8)
    res = torch.zeros(z.shape[0], outs[0].shape[1], oh, ow, dtype=outs[0].dtype, device=z.device)
    for o, ib, ob in zip(outs, in_b, out_b):
        padded = [i * 8 if is_decoder else i // 8 for i in ib]
        m = [ob[i] - padded[i] for i in range(4)]
        res[:, :, ob[2]:ob[3], ob[0]:ob[1]] = o[:, :, m[2]:o.size(2) + m[3], m[0]:o.size(3) + m[1]]
    return res


def adaptive_instance_normalization(content, style, eps=1e-5):
    """adaptive_instance_normalization + calc_mean_std (SUPIR/utils/colorfix.py:45-70): per (n, c) mean and UNBIASED variance
    (+eps) over the pixels; content is whitened with its own statistics and re-coloured with the style's."""
    n, c = content.shape[:2]

    def mean_std(t):
        v = t.reshape(n, c, -1)
        return v.mean(dim=2).view(n, c, 1, 1), (v.var(dim=2) + eps).sqrt().view(n, c, 1, 1)
    sm, ss = mean_std(style)
    cm, cs = mean_std(content)
    return (content - cm) / cs * ss + sm

