"""fp32 forms of the operators of supir_amd.ops, on libsupir_hip_f32.so (include/supir_hip_f32.h, csrc/f32/f32.hip).

What a `--diff_dtype fp32` / `--ae_dtype fp32` request runs on (reference: test.py:66-67; torch.autocast disables itself for float32,
sgm/modules/diffusionmodules/wrappers.py:87; SUPIR/models/SUPIR_model.py:41-69 -- the reference then computes in plain fp32).  Every
function here has the signature and layout contract of its namesake in ops.py (which dispatches to it by the operands' dtype), so the
module layer is the same code in all three element types; only the fusions that exist for speed (LayerNorm folded into GEMMs, fused q|k|v,
fused cross-attention, grouped launches, GroupNorm partials, autotuned tiles) are absent: an fp32 request is a correctness service.

Attention is batched GEMM -> row softmax -> batched GEMM with the fp32 scores in HBM (B * H * Tq * Tk_pad floats: 1.3 GB for SDXL's
largest self-attention of a CFG-doubled 1024^2 image; 1 GB for the VAE mid block at 1024^2).
"""
import ctypes as _ct
import math

import torch

from . import _lib

F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rows_ld(t):
    from .ops import _rows_ld as f
    return f(t)


def _check(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.SupirHipError("supir_amd ops need CUDA(HIP) tensors: the product path has no CPU fallback")
        assert t.dtype == F32, f"fp32 operator called with a {t.dtype} operand: element types are never mixed"


def _gemm_call(**kw):
    lib = _lib.load_f32()
    d = _lib.F32GemmDesc(nz0=1, nz1=1, alpha=1.0)
    for k, v in kw.items():
        setattr(d, k, v)
    _lib.check(lib.supir_f32_gemm(_ct.byref(d), _stream()), "supir_f32_gemm", lib)


def gemm(a, w, bias=None, *, rowbias=None, rows_per_batch=0, residual=None, act=0, alpha=1.0, out=None, out_dtype=None, tile=-1,
         alt16=None, gn_part=False):
    """ops.gemm in fp32.  act = 2 (GEGLU): `w` / `bias` are in the 32-row value / gate interleave of weights.interleave_geglu, as for the
    16-bit kernels; the projection is written to a scratch [M, N] and gated by supir_f32_geglu."""
    _check(a, w, bias, rowbias, residual, out)
    M, K, lda = _rows_ld(a)
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous() and out_dtype in (None, F32)
    n_out = N // 2 if act == 2 else N
    if out is None:
        out = torch.empty(*a.shape[:-1], n_out, dtype=F32, device=a.device)
    Mo, No, ldc = _rows_ld(out)
    assert Mo == M and No == n_out
    ldr = 0
    if residual is not None:
        Mr, Nr, ldr = _rows_ld(residual)
        assert Mr == M and Nr == n_out
    ld_rb = 0
    if rowbias is not None:
        assert rowbias.stride(-1) == 1 and rowbias.shape[-1] == N and rows_per_batch > 0
        ld_rb = rowbias.stride(0)
    if act == 2:
        assert residual is None and alpha == 1.0
        proj = torch.empty(M, N, dtype=F32, device=a.device)
        _gemm_call(A=a.data_ptr(), W=w.data_ptr(), C=proj.data_ptr(), bias=_p(bias), rowbias=_p(rowbias), kind=_lib.F32_GEMM, M=M, N=N, K=K,
                   lda=lda, ldw=K, ldc=N, ld_rowbias=ld_rb, rows_per_batch=rows_per_batch)
        lib = _lib.load_f32()
        _lib.check(lib.supir_f32_geglu(proj.data_ptr(), out.data_ptr(), M, N, N, ldc, 32, _stream()), "supir_f32_geglu", lib)
    else:
        _gemm_call(A=a.data_ptr(), W=w.data_ptr(), C=out.data_ptr(), bias=_p(bias), rowbias=_p(rowbias), residual=_p(residual),
                   kind=_lib.F32_GEMM, M=M, N=N, K=K, lda=lda, ldw=K, ldc=ldc, ldr=ldr, ld_rowbias=ld_rb, rows_per_batch=rows_per_batch,
                   act=act, alpha=alpha)
    return (out, None) if gn_part else out


def gemm_t(a, w, bias, B, T, Tpad, out=None, tile=-1):
    """ops.gemm_t in fp32: out[b][n][t] = a[b T + t] . w[n] (+ bias), [B, N, Tpad], zero padded."""
    _check(a, w, bias, out)
    M, K, lda = _rows_ld(a)
    N = w.shape[0]
    assert M == B * T and w.shape[1] == K and w.is_contiguous()
    if out is None:
        out = torch.zeros(B, N, Tpad, dtype=F32, device=a.device) if Tpad != T else torch.empty(B, N, Tpad, dtype=F32, device=a.device)
    assert out.shape == (B, N, Tpad) and out.is_contiguous()
    _gemm_call(A=a.data_ptr(), W=w.data_ptr(), C=out.data_ptr(), bias=_p(bias), kind=_lib.F32_GEMM, M=M, N=N, K=K, lda=lda, ldw=K, ldc=Tpad,
               rows_per_batch=T, out_mode=2)
    return out


def conv3x3(x, w, bias=None, *, stride=1, pad=(1, 1), upsample=False, out_hw=None, rowbias=None, residual=None, act=0, alpha=1.0, out=None,
            tile=-1, gn_part=False):
    """ops.conv3x3 in fp32: x [B,H,W,Cin(ld)] -> [B,OH,OW,Cout]; w [Cout,3,3,Cin]."""
    _check(x, w, bias, rowbias, residual, out)
    B, H, W, Cin = x.shape
    _, _, ldx = _rows_ld(x)
    Cout = w.shape[0]
    assert w.shape[1:] == (3, 3, Cin) and w.is_contiguous()
    if out_hw is None:
        if upsample:
            out_hw = (2 * H, 2 * W)
        elif stride == 1:
            out_hw = (H, W)
        else:
            out_hw = ((H + 2 * pad[0] - 3) // stride + 1, (W + 2 * pad[1] - 3) // stride + 1)
    OH, OW = out_hw
    if out is None:
        out = torch.empty(B, OH, OW, Cout, dtype=F32, device=x.device)
    _, _, ldy = _rows_ld(out)
    ldr = 0
    if residual is not None:
        assert residual.shape == out.shape
        _, _, ldr = _rows_ld(residual)
    ld_rb = 0
    if rowbias is not None:
        assert rowbias.shape == (B, Cout) and rowbias.stride(-1) == 1
        ld_rb = rowbias.stride(0)
    _gemm_call(A=x.data_ptr(), W=w.data_ptr(), C=out.data_ptr(), bias=_p(bias), rowbias=_p(rowbias), residual=_p(residual),
               kind=_lib.F32_CONV3X3, M=B * OH * OW, N=Cout, K=9 * Cin, lda=ldx, ldw=9 * Cin, ldc=ldy, ldr=ldr, ld_rowbias=ld_rb,
               rows_per_batch=OH * OW, act=act, alpha=alpha, B=B, H=H, Wd=W, Cin=Cin, OH=OH, OW=OW, stride=stride, pad_t=pad[0], pad_l=pad[1],
               upsample=1 if upsample else 0)
    return (out, None) if gn_part else out


def softmax_rows(s, scale, out=None, valid=None, dtype=None):
    """softmax over the first `valid` columns of fp32 scores [rows, Tpad]; the remaining columns of the output are zero."""
    _check(s, out)
    lib = _lib.load_f32()
    rows, Tp = s.shape
    T = Tp if valid is None else valid
    assert s.stride(1) == 1
    if out is None:
        out = torch.empty(rows, Tp, dtype=F32, device=s.device)
    _lib.check(lib.supir_f32_softmax_rows(s.data_ptr(), out.data_ptr(), rows, T, Tp, s.stride(0), out.stride(0), scale, 0, _stream()),
               "supir_f32_softmax_rows", lib)
    return out


def _attention(q, k, vt, B, H, D, Tq, Tk, ldq, ldk, ldvt, out, ldo, scale, causal=False):
    """softmax(scale q_h k_h^T) v_h for every (batch, head): scores [B, H, Tq, Tkp] in HBM, three launches.
    causal (text towers): query i sees keys j <= i -- the mask is applied by the softmax (masked probabilities are exact zeros)."""
    assert not causal or Tq == Tk
    lib = _lib.load_f32()
    Tkp = ldvt
    s = torch.empty(B, H, Tq, Tkp, dtype=F32, device=q.device)
    # S[b, h] = q[b, :, h D:(h + 1) D] . k[b, :, h D:(h + 1) D]^T   (z0 = head, z1 = batch element)
    _gemm_call(A=q.data_ptr(), W=k.data_ptr(), C=s.data_ptr(), kind=_lib.F32_GEMM, M=Tq, N=Tk, K=D, lda=ldq, ldw=ldk, ldc=Tkp, nz0=H, nz1=B,
               a_s0=D, a_s1=Tq * ldq, w_s0=D, w_s1=Tk * ldk, c_s0=Tq * Tkp, c_s1=H * Tq * Tkp)
    _lib.check(lib.supir_f32_softmax_rows(s.data_ptr(), s.data_ptr(), B * H * Tq, Tk, Tkp, Tkp, Tkp, scale, Tq if causal else 0, _stream()),
               "supir_f32_softmax_rows", lib)
    # O[b, :, h D:(h + 1) D] = P[b, h] . (V^T[b, h D:(h + 1) D, :])^T   (padding columns: P is zero there, V^T is zero there)
    _gemm_call(A=s.data_ptr(), W=vt.data_ptr(), C=out.data_ptr(), kind=_lib.F32_GEMM, M=Tq, N=D, K=Tkp, lda=Tkp, ldw=ldvt, ldc=ldo, nz0=H, nz1=B,
               a_s0=Tq * Tkp, a_s1=H * Tq * Tkp, w_s0=D * ldvt, w_s1=H * D * ldvt, c_s0=D, c_s1=Tq * ldo)
    return out


def flash_attn(q, k, vt, B, H, Tq, Tk, out=None, causal=False):
    """ops.flash_attn in fp32: q [B,Tq,>=H*64] k [B,Tk,>=H*64] (row-strided views), vt [B,H*64,Tpad] zero padded -> [B,Tq,H*64]."""
    _check(q, k, vt, out)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and vt.is_contiguous()
    ldq, ldk, ldvt = q.stride(-2), k.stride(-2), vt.shape[-1]
    assert q.shape[0] == B and q.stride(0) == Tq * ldq and k.stride(0) == Tk * ldk and ldvt >= Tk
    if out is None:
        out = torch.empty(B, Tq, H * 64, dtype=F32, device=q.device)
    return _attention(q, k, vt, B, H, 64, Tq, Tk, ldq, ldk, ldvt, out, out.stride(-2), 0.125, causal)


def flash_attn_d512(q, k, vt, Tk, out=None, splits=0):
    """ops.flash_attn_d512 in fp32 (VAE mid block, one head of 512 channels)."""
    _check(q, k, vt, out)
    B, Tq, C = q.shape
    assert C == 512 and k.shape == (B, Tk, 512) and vt.shape[:2] == (B, 512) and vt.is_contiguous()
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and q.stride(0) == Tq * q.stride(1) and k.stride(0) == Tk * k.stride(1)
    if out is None:
        out = torch.empty(B, Tq, 512, dtype=F32, device=q.device)
    return _attention(q, k, vt, B, 1, 512, Tq, Tk, q.stride(1), k.stride(1), vt.shape[-1], out, out.stride(1), 512 ** -0.5)


_WS = {}


def _gn_ws(B, device):
    key = (B, device, _stream())       # per stream: GroupNorms of the two branches may run concurrently
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(B * 32 * 64 * 2 + B * 32 * 2, dtype=torch.float64, device=device)
    return ws


def groupnorm(x, gamma, beta, eps, *, silu=False, x2=None, mod_g=None, mod_b=None, control_scale=1.0, x1raw=None, x2raw=None, out=None,
              given=None, part=None, part2=None):
    """ops.groupnorm in fp32 (statistics in fp64; GroupNorm partials of a producer are a 16-bit-path fusion and are ignored).
    given: fp32 [B, 32, 2] externally pooled (mean, biased variance) -- the tiled VAE (utils/tilevae.py pooled_groupnorm)."""
    _check(x, gamma, beta, x2, mod_g, mod_b, x1raw, x2raw, out, given)
    assert given is None or (given.shape == (x.shape[0], 32, 2) and given.is_contiguous())
    lib = _lib.load_f32()
    B = x.shape[0]
    HW = int(math.prod(x.shape[1:-1]))
    _, C1, ld1 = _rows_ld(x)
    C, ld2 = C1, 0
    if x2 is not None:
        _, C2, ld2 = _rows_ld(x2)
        assert x2.shape[:-1] == x.shape[:-1]
        C = C1 + C2
    assert gamma.numel() == C
    if out is None:
        out = torch.empty(*x.shape[:-1], C, dtype=F32, device=x.device)
    _, Co, ldo = _rows_ld(out)
    assert Co == C
    ldm = 0
    if mod_g is not None:
        _, Cm, ldm = _rows_ld(mod_g)
        _, Cm2, ldm2 = _rows_ld(mod_b)
        assert Cm == C and Cm2 == C and ldm2 == ldm
    for raw, src in ((x1raw, x), (x2raw, x2)):
        if raw is not None:
            assert raw.shape == src.shape and _rows_ld(raw)[2] == _rows_ld(src)[2]
    ws = _gn_ws(B, x.device)
    rc = lib.supir_f32_groupnorm(x.data_ptr(), _p(x2), _p(x1raw), _p(x2raw), B, HW, C, C1, ld1, ld2, gamma.data_ptr(), beta.data_ptr(), eps,
                                 1 if silu else 0, _p(mod_g), _p(mod_b), ldm, float(control_scale), out.data_ptr(), ldo, ws.data_ptr(),
                                 ws.numel() * 8, _p(given), _stream())
    _lib.check(rc, "supir_f32_groupnorm", lib)
    return out


def groupnorm_stats(x):
    """(sum, sum of squares) per (batch, group) of a channels-last fp32 tensor: FP64 [B, 32, 2] (utils/tilevae.py pool_statistics works in
    fp64 either way; the 16-bit path hands it fp32 sums)."""
    _check(x)
    lib = _lib.load_f32()
    B = x.shape[0]
    HW = int(math.prod(x.shape[1:-1]))
    _, C, ld = _rows_ld(x)
    ws = _gn_ws(B, x.device)
    out = torch.empty(B, 32, 2, dtype=torch.float64, device=x.device)
    _lib.check(lib.supir_f32_groupnorm_stats(x.data_ptr(), B, HW, C, ld, out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()),
               "supir_f32_groupnorm_stats", lib)
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _check(x, gamma, beta, out)
    lib = _lib.load_f32()
    rows, C, ldx = _rows_ld(x)
    if out is None:
        out = torch.empty(*x.shape, dtype=F32, device=x.device)
    _, _, ldy = _rows_ld(out)
    _lib.check(lib.supir_f32_layernorm(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, C, ldx, ldy, eps, _stream()),
               "supir_f32_layernorm", lib)
    return out


def conv3x3_smallcin(x_nchw, w, bias, add=None, out=None, dtype=None):
    """fp32 NCHW [B,Cin<=8,H,W] -> fp32 NHWC [B,H,W,Cout] (+ add); w fp32 [Cout,Cin,3,3] (the reference layout).  The input is put into
    NHWC by a torch copy (a 3- / 4-channel image or latent), the product is the general kernel with scalar loads (Cin % 4 != 0 allowed)."""
    _check(x_nchw, w, bias, add, out)
    xh = x_nchw.permute(0, 2, 3, 1).contiguous()
    wk = w.permute(0, 2, 3, 1).contiguous()
    return conv3x3(xh, wk, bias, residual=add, out=out)


def conv3x3_smallcout(x, w9, bias, out=None):
    """fp32 NHWC [B,H,W,Cin] -> fp32 NCHW [B,Cout<=8,H,W]; w9 [9,Cout,Cin] (weights.conv3x3_w9)."""
    _check(x, w9, bias, out)
    Cout, Cin = w9.shape[1], w9.shape[2]
    wk = w9.permute(1, 0, 2).reshape(Cout, 3, 3, Cin).contiguous()
    y = conv3x3(x, wk, bias).permute(0, 3, 1, 2)
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out
