"""Image I/O edges of test.py on the GPU: `PIL2Tensor` / `Tensor2PIL` with the reference's signatures and
results (SUPIR/util.py:60-94).

PIL2Tensor: the reference resizes on the CPU with PIL's `Image.resize(BICUBIC)` (Pillow's separable 8-bit resampler, 22-bit
fixed-point coefficients) and converts to a float tensor.  Here the decoded uint8 pixels go to the device once; two launches of
`supir_resample_u8` (horizontal, then vertical -- Pillow's order) reproduce Pillow's integer arithmetic bit for bit, and the
second pass writes the fp32 CHW tensor in [-1, 1] directly through a 256-entry table evaluated exactly as numpy evaluates
`x / 255 * 2 - 1`.  The coefficient tables are rebuilt here in float64 with Pillow's formulas (precompute_coeffs /
normalize_coeffs_8bpc, src/libImaging/Resample.c -- third party, Pillow >= 7; operation order kept so the rounded integers agree).

Tensor2PIL: `F.interpolate(mode='bicubic')` + `* 127.5 + 127.5` + clip + uint8 truncation in one launch of `supir_bicubic_f32`.
"""
import math

import numpy as np
import torch

from .. import _lib

_PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_bicubic_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the full-image box (in0 = 0, in1 = in_size):
    returns (bounds int32 [out, 2], kk int32 [out, ksize], ksize)."""
    support_f = 2.0
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support_f * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww = 0.0
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = _bicubic_filter((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
    return bounds, kk, ksize


_LUT = None


def _lut(device):
    """x / 255 * 2 - 1 exactly as SUPIR/util.py:81-82 evaluates it (numpy float64, then one rounding to float32)."""
    global _LUT
    if _LUT is None or _LUT.device != torch.device(device):
        x = np.arange(256, dtype=np.uint8)
        _LUT = torch.tensor(x / 255 * 2 - 1, dtype=torch.float32).to(device)
    return _LUT


def target_size(w, h, upsacle=1, min_size=1024, fix_resize=None):
    """The size arithmetic of PIL2Tensor (SUPIR/util.py:65-78): returns (w, h, w0, h0)."""
    w *= upsacle
    h *= upsacle
    w0, h0 = round(w), round(h)
    if min(w, h) < min_size:
        _u = min_size / min(w, h)
        w *= _u
        h *= _u
    if fix_resize is not None:
        _u = fix_resize / min(w, h)
        w *= _u
        h *= _u
        w0, h0 = round(w), round(h)
    w = int(np.round(w / 64.0)) * 64
    h = int(np.round(h / 64.0)) * 64
    return w, h, w0, h0


def resize_bicubic_u8(img_u8, out_w, out_h, want_u8=False):
    """img_u8: uint8 [H, W, C] on the device -> fp32 [C, out_h, out_w] in [-1, 1] (and the resized uint8 image if asked)."""
    lib = _lib.load()
    if not img_u8.is_cuda:
        raise _lib.SupirHipError("supir_amd image I/O needs a CUDA(HIP) tensor: the product path has no CPU fallback")
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.is_contiguous()
    H, W, C = img_u8.shape
    dev = img_u8.device
    st = torch.cuda.current_stream().cuda_stream
    cur, cw = img_u8, W
    if out_w != W:      # Pillow: horizontal pass first
        b, k, ks = pillow_bicubic_coeffs(W, out_w)
        bt, kt = torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev)
        tmp = torch.empty(H, out_w, C, dtype=torch.uint8, device=dev)
        _lib.check(lib.supir_resample_u8(cur.data_ptr(), tmp.data_ptr(), None, None, bt.data_ptr(), kt.data_ptr(), ks, H, cw, H, out_w,
                                         C, 0, st), "supir_resample_u8(h)")
        cur, cw = tmp, out_w
    out_f = torch.empty(C, out_h, out_w, dtype=torch.float32, device=dev)
    out_u = torch.empty(out_h, out_w, C, dtype=torch.uint8, device=dev) if want_u8 else None
    if out_h != H:
        b, k, ks = pillow_bicubic_coeffs(H, out_h)
    else:               # identity vertical pass: one tap of weight 1 (keeps a single code path for the fp32 / CHW conversion)
        b = np.stack([np.arange(out_h), np.ones(out_h)], 1).astype(np.int32)
        k = np.full((out_h, 1), 1 << _PRECISION_BITS, dtype=np.int32)
        ks = 1
    bt, kt = torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev)
    _lib.check(lib.supir_resample_u8(cur.data_ptr(), 0 if out_u is None else out_u.data_ptr(), out_f.data_ptr(), _lut(dev).data_ptr(),
                                     bt.data_ptr(), kt.data_ptr(), ks, H, cw, out_h, out_w, C, 1, st), "supir_resample_u8(v)")
    return (out_f, out_u) if want_u8 else out_f


def unpremultiply_u8(u8):
    """uint8 [H, W, C + 1] with premultiplied colour bands ("RGBa" / "La") -> straight alpha, Pillow's rgba2rgbA / la2lA
    (src/libImaging/Convert.c): c' = min(255, 255 * c // alpha), colour passed through where alpha is 0 or 255."""
    c, a = u8[..., :-1].to(torch.int32), u8[..., -1:].to(torch.int32)
    un = torch.where((a == 0) | (a == 255), c, ((255 * c) // a.clamp(min=1)).clamp(max=255))
    return torch.cat([un, a], dim=-1).to(torch.uint8)


def PIL2Tensor(img, upsacle=1, min_size=1024, fix_resize=None, device="cuda"):
    """PIL.Image -> (Tensor[C, H, W] RGB in [-1, 1] on `device`, h0, w0): SUPIR/util.py:60-83 (argument names as there)."""
    w, h = img.size
    w, h, w0, h0 = target_size(w, h, upsacle, min_size, fix_resize)
    if img.mode in ("RGBA", "LA"):
        # Pillow resamples images with an alpha band in PREMULTIPLIED form (Image.resize: convert to "RGBa" / "La", resample every
        # band, convert back), so the colour bands the reference gets differ from a per-band resample wherever alpha < 255.  The
        # two conversions are 8-bit per-pixel arithmetic: the premultiply is Pillow's own convert() on the decoded image, the
        # un-premultiply (Convert.c rgba2rgbA: CLIP8(255 * c / alpha), pass-through for alpha 0 / 255) is integer torch ops.
        pre = np.asarray(img.convert("RGBa" if img.mode == "RGBA" else "La"))
        _, u8 = resize_bicubic_u8(torch.from_numpy(np.array(pre)).to(device), w, h, want_u8=True)
        u8 = unpremultiply_u8(u8)
        return _lut(u8.device)[u8.long()].permute(2, 0, 1).contiguous(), h0, w0
    if img.mode not in ("L", "RGB"):
        raise ValueError(f"PIL2Tensor: image mode {img.mode!r} is not supported (convert to RGB / RGBA / L / LA first): Pillow "
                         "resamples palette, 16-bit and float modes through other code paths than the 8-bit kernel reproduces")
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    x = resize_bicubic_u8(torch.from_numpy(np.array(arr)).to(device), w, h)   # np.array: a writable copy (PIL hands out read-only views)
    return x, h0, w0


def bicubic_resize_f32(x, h0, w0, want_u8=True):
    """x fp32 [C, H, W] on the device -> (uint8 [h0, w0, C] of Tensor2PIL, fp32 [C, h0, w0] of F.interpolate bicubic)."""
    lib = _lib.load()
    if not x.is_cuda:
        raise _lib.SupirHipError("supir_amd image I/O needs a CUDA(HIP) tensor: the product path has no CPU fallback")
    x = x.float().contiguous()
    C, H, W = x.shape
    out_u = torch.empty(h0, w0, C, dtype=torch.uint8, device=x.device) if want_u8 else None
    out_f = torch.empty(C, h0, w0, dtype=torch.float32, device=x.device)
    _lib.check(lib.supir_bicubic_f32(x.data_ptr(), 0 if out_u is None else out_u.data_ptr(), out_f.data_ptr(), C, H, W, h0, w0,
                                     torch.cuda.current_stream().cuda_stream), "supir_bicubic_f32")
    return out_u, out_f


def Tensor2PIL(x, h0, w0):
    """Tensor[C, H, W] RGB in [-1, 1] -> PIL.Image of size (w0, h0): SUPIR/util.py:86-94."""
    from PIL import Image
    out_u, _ = bicubic_resize_f32(x, h0, w0)
    return Image.fromarray(out_u.cpu().numpy())
