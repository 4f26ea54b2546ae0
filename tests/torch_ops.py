"""TEST-ONLY stand-in for EVERY `supir_amd.ops` entry point the module layer calls, in plain torch, so that the HOST logic of the
whole path -- derived weight layouts (supir_amd/weights.py: LayerNorm folds, GEGLU interleave, fused q|k|v stacks, concatenated
gamma|beta / embedding matrices), channel-slice views, in-place residual streams, caches, ControlWrapper / SUPIRModel /
samplers / conditioner / tiled VAE plumbing -- runs on a box without a GPU and can be held to the reference's goldens and to the
oracle in the `-m "not gpu"` tier.  Same signatures, operand layouts and return conventions as the real ops (channels-last
activations, K-contiguous weights, V^T buffers, RowStats row statistics); arithmetic in fp32 on whatever device the operands
live on, outputs rounded to the operand dtype.

Never imported by the product (`grep -rn torch_ops supir_amd/` is empty): supir_amd.ops itself refuses non-CUDA tensors.  On the GPU
the same functions double as a per-op reference for the HIP kernels (tests/test_torch_ops_vs_hip_gpu.py), which is what pins
THIS file's reading of the layouts to the kernels'."""
import contextlib
import math

import torch
import torch.nn.functional as F

from supir_amd import ops as real_ops
from supir_amd import weights as Wt

PREFER_FUSED_QKV = True       # what `choose` answers for the fused-vs-separate q|k|v question (tests flip it)
GnPart = real_ops.GnPart
_k = real_ops._k
USE_GEMM16 = real_ops.USE_GEMM16


class RowStats:
    """Same role as ops.RowStats: one slot of (sum, sum of squares) per row, fp32 [M, 2, 2] (ld = 2: the real buffers keep ld even)."""
    __slots__ = ("buf", "slots", "ld")

    def __init__(self, buf, slots, ld):
        self.buf, self.slots, self.ld = buf, slots, ld


def _act(y, act):
    if act == 0:
        return y
    if act == 1:
        return F.silu(y)
    if act == 2:      # GEGLU, W rows interleaved [32 value | 32 gate] per 64 (include/supir_hip.h SUPIR_ACT_GEGLU)
        n = y.shape[-1]
        v = y.reshape(*y.shape[:-1], n // 64, 2, 32)
        return (v[..., 0, :] * F.gelu(v[..., 1, :])).reshape(*y.shape[:-1], n // 2)
    if act == 3:
        return F.gelu(y)
    if act == 4:
        return y * torch.sigmoid(1.702 * y)
    raise ValueError(act)


def _store(y, out, shape, dtype):
    """Round to the output type; write into `out` when given (in-place residual streams alias `residual`)."""
    if out is None:
        return y.to(dtype).reshape(shape)
    out.copy_(y.reshape(out.shape).to(out.dtype))
    return out


def _rowbias_rows(rowbias, M, rows_per_batch):
    nb = rowbias.shape[0]
    rpb = rows_per_batch if rows_per_batch > 0 else M // nb
    return rowbias.float().repeat_interleave(rpb, dim=0)[:M]


def gemm(a, w, bias=None, *, rowbias=None, rows_per_batch=0, residual=None, act=0, alpha=1.0, out=None, out_dtype=None, tile=-1,
         alt16=None, gn_part=False):
    DT = a.dtype
    K = a.shape[-1]
    assert w.shape[1] == K and w.dtype == DT
    a2 = a.reshape(-1, K).float()
    y = a2 @ w.float().T
    if bias is not None:
        y = y + bias.float()
    if rowbias is not None:
        y = y + _rowbias_rows(rowbias, a2.shape[0], rows_per_batch)
    y = alpha * _act(y, act)
    if residual is not None:
        y = y + residual.reshape(-1, y.shape[-1]).float()
    o = _store(y, out, (*a.shape[:-1], y.shape[-1]), DT if out_dtype is None else out_dtype)
    return (o, None) if gn_part else o


def _ln_fold(y, a2, ln, colsum, K, eps):
    """supir_gemm_bf16_ln consumer half: rstd * (x.W' - mean * colsum)."""
    if ln.slots == 0:
        mean, rstd = ln.buf[:, 0], ln.buf[:, 1]
    else:
        s = ln.buf[:, :ln.slots].sum(dim=1)
        mean = s[:, 0] / K
        rstd = torch.rsqrt((s[:, 1] / K - mean * mean).clamp_min(0.0) + eps)
    return rstd[:, None] * (y - mean[:, None] * colsum.float()[None, :])


def _emit_stats(vals):
    """Row statistics of the STORED (rounded) values, one slot."""
    v = vals.reshape(-1, vals.shape[-1]).float()
    buf = torch.zeros(v.shape[0], 2, 2, dtype=torch.float32, device=v.device)
    buf[:, 0, 0], buf[:, 0, 1] = v.sum(dim=1), (v * v).sum(dim=1)
    return RowStats(buf, 1, 2)


def _transposed(y, B, T, Tpad, DT, out):
    N = y.shape[-1]
    yt = y.reshape(B, T, N).transpose(1, 2).to(DT)
    if out is None:
        out = torch.zeros(B, N, Tpad, dtype=DT, device=y.device)
    out[:, :, :T] = yt
    return out


def gemm_ln(a, w, bias=None, *, residual=None, act=0, alpha=1.0, out=None, tile=-1, emit_stats=False, ln=None, colsum=None,
            ln_eps=1e-5, trans=None, alt16=None):
    DT = a.dtype
    K = a.shape[-1]
    a2 = a.reshape(-1, K).float()
    y = a2 @ w.float().T
    if ln is not None:
        assert colsum is not None and colsum.numel() == w.shape[0]
        y = _ln_fold(y, a2, ln, colsum, K, ln_eps)
    if bias is not None:
        y = y + bias.float()
    y = alpha * _act(y, act)
    if residual is not None:
        y = y + residual.reshape(-1, y.shape[-1]).float()
    if trans is not None:
        B, T, Tpad = trans
        return _transposed(y, B, T, Tpad, DT, out)
    o = _store(y, out, (*a.shape[:-1], y.shape[-1]), DT)
    return (o, _emit_stats(o)) if emit_stats else o


gemm_qkv_supported = real_ops.gemm_qkv_supported


def gemm_qkv(a, w, bias, B, T, n_split, *, ln=None, colsum=None, ln_eps=1e-5, out_qk=None, out_vt=None):
    DT = a.dtype
    K = a.shape[-1]
    a2 = a.reshape(-1, K).float()
    y = a2 @ w.float().T
    if ln is not None:
        y = _ln_fold(y, a2, ln, colsum, K, ln_eps)
    if bias is not None:
        y = y + bias.float()
    qk = _store(y[:, :n_split], out_qk, (B, T, n_split), DT)
    return qk, _transposed(y[:, n_split:], B, T, T, DT, out_vt)


def has_fused(dtype):
    return True


def choose(key, fns, prefer=None, margin=0.1):
    return prefer if (prefer is not None and PREFER_FUSED_QKV) else 0


def gemm_t(a, w, bias, B, T, Tpad, out=None, tile=-1):
    y = a.reshape(-1, a.shape[-1]).float() @ w.float().T
    if bias is not None:
        y = y + bias.float()
    return _transposed(y, B, T, Tpad, a.dtype, out)


def rowstats_finalize(st, dim, eps):
    if st.slots == 0:
        return st
    s = st.buf[:, :st.slots].sum(dim=1)
    mean = s[:, 0] / dim
    rstd = torch.rsqrt((s[:, 1] / dim - mean * mean).clamp_min(0.0) + eps)
    return RowStats(torch.stack([mean, rstd], dim=1), 0, 0)


def conv3x3(x, w, bias=None, *, stride=1, pad=(1, 1), upsample=False, out_hw=None, rowbias=None, residual=None, act=0, alpha=1.0,
            out=None, tile=-1, gn_part=False):
    """x [B,H,W,Cin] channels-last (a channel slice of a wider buffer is fine), w [Cout,3,3,Cin]."""
    DT = x.dtype
    xin = x.float().permute(0, 3, 1, 2)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    wt = w.float().permute(0, 3, 1, 2)
    b = None if bias is None else bias.float()
    if stride == 2 and tuple(pad) == (0, 0):     # VAE downsample: taps beyond the bottom / right edge read zero (model.py:81-86)
        y = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wt, b, stride=2)
    else:
        y = F.conv2d(xin, wt, b, stride=stride, padding=1)
    y = y.permute(0, 2, 3, 1)
    if out_hw is not None:
        assert tuple(y.shape[1:3]) == tuple(out_hw), (y.shape, out_hw)
    if rowbias is not None:
        y = y + rowbias.float()[:, None, None, :]
    y = alpha * _act(y, act)
    if residual is not None:
        y = y + residual.float()
    o = _store(y.contiguous(), out, tuple(y.shape), DT)
    return (o, None) if gn_part else o


def flash_attn(q, k, vt, B, H, Tq, Tk, out=None, causal=False):
    inner = H * 64
    qh = q[..., :inner].float().reshape(B, Tq, H, 64).permute(0, 2, 1, 3)
    kh = k[..., :inner].float().reshape(B, Tk, H, 64).permute(0, 2, 1, 3)
    vh = vt[:, :, :Tk].float().reshape(B, H, 64, Tk).transpose(-1, -2)
    o = F.scaled_dot_product_attention(qh, kh, vh, is_causal=causal, scale=0.125)
    o = o.permute(0, 2, 1, 3).reshape(B, Tq, inner)
    return _store(o, out, (B, Tq, inner), q.dtype)


xattn_q_supported = real_ops.xattn_q_supported


def xattn_q(x, wq, bias, k, vt, B, H, T, Tk, *, ln=None, colsum=None, ln_eps=1e-5, out=None):
    q = gemm_ln(x, wq, bias, ln=ln, colsum=colsum, ln_eps=ln_eps)
    return flash_attn(q.reshape(B, T, H * 64), k, vt, B, H, T, Tk, out=out)


use_flash_d512 = real_ops.use_flash_d512


def flash_attn_d512(q, k, vt, Tk, out=None, splits=0):   # key splits change the summation order only
    B, Tq, C = q.shape
    v = vt[:, :, :Tk].float().transpose(1, 2)
    o = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v[:, None], scale=C ** -0.5)[:, 0]
    return _store(o, out, (B, Tq, C), q.dtype)


def softmax_rows(s, scale, out=None, valid=None, dtype=None):
    rows, Tp = s.shape
    T = Tp if valid is None else valid
    DT = out.dtype if out is not None else (Wt.cdt() if dtype is None else dtype)
    p = torch.zeros(rows, Tp, dtype=torch.float32, device=s.device)
    p[:, :T] = torch.softmax(s[:, :T].float() * scale, dim=-1)
    return _store(p, out, (rows, Tp), DT)


def groupnorm_stats(x):
    B, C = x.shape[0], x.shape[-1]
    g = x.float().reshape(B, -1, 32, C // 32)
    return torch.stack([g.sum(dim=(1, 3)), (g * g).sum(dim=(1, 3))], dim=-1)


def groupnorm(x, gamma, beta, eps, *, silu=False, x2=None, mod_g=None, mod_b=None, control_scale=1.0, x1raw=None, x2raw=None,
              out=None, given=None, part=None, part2=None):
    DT = x.dtype
    cat = x.float() if x2 is None else torch.cat([x.float(), x2.float()], dim=-1)
    B, C = cat.shape[0], cat.shape[-1]
    lead = cat.shape[:-1]
    g = cat.reshape(B, -1, 32, C // 32)
    if given is None:
        mean = g.mean(dim=(1, 3), keepdim=True)
        var = (g * g).mean(dim=(1, 3), keepdim=True) - mean * mean
    else:
        mean, var = given[..., 0].float().view(B, 1, 32, 1), given[..., 1].float().view(B, 1, 32, 1)
    y = ((g - mean) * torch.rsqrt(var.clamp_min(0.0) + eps)).reshape(*lead, C) * gamma.float() + beta.float()
    if mod_g is not None:
        y = y * (mod_g.float() + 1.0) + mod_b.float()
    if silu:
        y = F.silu(y)
    if float(control_scale) != 1.0:
        r1 = (x if x1raw is None else x1raw).float()
        raw = r1 if x2 is None else torch.cat([r1, (x2 if x2raw is None else x2raw).float()], dim=-1)
        y = y * control_scale + raw * (1.0 - control_scale)
    return _store(y, out, tuple(y.shape), DT)


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    return _store(y, out, tuple(x.shape), x.dtype)


def conv3x3_smallcin(x_nchw, w, bias, add=None, out=None, dtype=None):
    DT = out.dtype if out is not None else add.dtype if add is not None else (Wt.cdt() if dtype is None else dtype)
    y = F.conv2d(x_nchw.float(), w.float(), None if bias is None else bias.float(), padding=1).permute(0, 2, 3, 1)
    if add is not None:
        y = y + add.float()
    return _store(y.contiguous(), out, tuple(y.shape), DT)


def conv3x3_smallcout(x, w9, bias, out=None):
    co, ci = w9.shape[1], w9.shape[2]
    w = w9.float().reshape(3, 3, co, ci).permute(2, 3, 0, 1)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None if bias is None else bias.float(), padding=1)
    return _store(y.contiguous(), out, tuple(y.shape), torch.float32)


def pointwise_nchw(x, w, bias, in_scale=1.0):
    co, ci = w.shape[:2]
    return F.conv2d(x.float() * in_scale, w.float().reshape(co, ci, 1, 1), None if bias is None else bias.float())


def edm_step_pre(x, eps, s_noise, noise_mul, c_in, reps):
    x_hat = x if eps is None else x + (eps * s_noise) * noise_mul
    return x_hat, torch.cat([x_hat * c_in] * reps, dim=0)


def edm_step_post(net_out, x_hat, x_center, c_out, c_skip, cfg_scale, restore_mul, sigma_hat, dt, reps):
    n = x_hat.shape[0]
    den = net_out[:n] * c_out + x_hat * c_skip
    if reps == 2:
        den1 = net_out[n:] * c_out + x_hat * c_skip
        den = den + cfg_scale * (den1 - den)
    if x_center is not None:
        den = den - (den - x_center.float()) * restore_mul
    return x_hat + dt * ((x_hat - den) / sigma_hat)


def wavelet_decomposition(img, levels=5, want_high=True):
    """colorfix.py:73-107 restated: dilated depthwise 3x3 blur with replicate padding, high = sum of (img_i - low_i)."""
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]], device=img.device) / 16.0
    cur = img.float()
    C = cur.shape[1]
    high = torch.zeros_like(cur) if want_high else None
    for i in range(levels):
        r = 2 ** i
        low = F.conv2d(F.pad(cur, (r, r, r, r), mode="replicate"), k[None, None].repeat(C, 1, 1, 1), groups=C, dilation=r)
        if want_high:
            high = high + (cur - low)
        cur = low
    return high, cur


class WeightPrefetch:
    """No caches to warm on this backend: the recording / replay protocol is accepted and ignored."""

    def __init__(self, distance=1, kind="inline"):
        self.distance, self.kind, self.mode = distance, kind, None

    def begin_record(self):
        self.mode = "record"

    def begin_replay(self, device):
        self.mode = "replay"

    def end(self):
        self.mode = None


def set_prefetch(pf):
    pass


def paired_run(fn_a, fn_b, side=None):
    return fn_a(), fn_b()


def start_trace(timed=False):
    return []


def stop_trace():
    return None


# ------------------------------------------------------------------------------------------------ image I/O edges
def resize_bicubic_u8(img_u8, out_w, out_h, want_u8=False):
    """supir_amd.utils.imageio.resize_bicubic_u8 through Pillow itself (the kernel reproduces Pillow bit for bit on the GPU)."""
    import numpy as np
    from PIL import Image
    arr = img_u8.cpu().numpy()
    bands = [np.asarray(Image.fromarray(arr[..., c]).resize((out_w, out_h), Image.BICUBIC)) for c in range(arr.shape[-1])]
    u8 = torch.from_numpy(np.stack(bands, axis=-1)).to(img_u8.device)
    lut = torch.tensor(np.arange(256, dtype=np.uint8) / 255 * 2 - 1, dtype=torch.float32, device=img_u8.device)
    f = lut[u8.long()].permute(2, 0, 1).contiguous()
    return (f, u8) if want_u8 else f


def bicubic_resize_f32(x, h0, w0, want_u8=True):
    """Tensor2PIL's arithmetic (SUPIR/util.py:86-94) with torch ops."""
    f = F.interpolate(x.float()[None], size=(h0, w0), mode="bicubic")[0]
    u8 = (f.permute(1, 2, 0) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8) if want_u8 else None
    return u8, f


_NAMES = ["gemm", "gemm_ln", "gemm_qkv", "gemm_qkv_supported", "choose", "gemm_t", "rowstats_finalize", "conv3x3", "flash_attn",
          "xattn_q", "xattn_q_supported", "use_flash_d512", "flash_attn_d512", "softmax_rows", "groupnorm_stats", "groupnorm", "layernorm", "conv3x3_smallcin",
          "conv3x3_smallcout", "pointwise_nchw", "edm_step_pre", "edm_step_post", "wavelet_decomposition", "WeightPrefetch",
          "set_prefetch", "paired_run", "start_trace", "stop_trace", "RowStats", "has_fused"]


@contextlib.contextmanager
def installed(fp32=True):
    """Swap the entry points of `supir_amd.ops` (and the two image-I/O kernels' wrappers) for this backend; fp32=True also makes
    fp32 the "compute dtype" of every scope (derived weights and activations stay fp32: results comparable with the fp32 oracle /
    the reference goldens to ~1e-5), fp32=False keeps bf16 / fp16 rounding of every stored tensor.  Restores everything on exit."""
    import sys
    from supir_amd.utils import imageio
    me = sys.modules[__name__]
    if real_ops.gemm is gemm:      # nested use: the outer scope owns the swap
        yield me
        return
    saved = {n: getattr(real_ops, n) for n in _NAMES}
    saved_io = (imageio.resize_bicubic_u8, imageio.bicubic_resize_f32)
    saved_cdt = (list(Wt._CDT), Wt.as_compute_dtype)
    try:
        for n in _NAMES:
            setattr(real_ops, n, getattr(me, n))
        imageio.resize_bicubic_u8, imageio.bicubic_resize_f32 = resize_bicubic_u8, bicubic_resize_f32
        if fp32:
            Wt._CDT[:] = [torch.float32]
            Wt.as_compute_dtype = lambda dtype: torch.float32
        yield me
    finally:
        for n, v in saved.items():
            setattr(real_ops, n, v)
        imageio.resize_bicubic_u8, imageio.bicubic_resize_f32 = saved_io
        Wt._CDT[:] = saved_cdt[0]
        Wt.as_compute_dtype = saved_cdt[1]
