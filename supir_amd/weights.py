"""One-time conversion of reference-layout fp32 parameters into the kernel layouts (bf16, K-contiguous).

The state-dict keys/shapes stay exactly the reference's (drop-in `load_state_dict`); these helpers build the derived
device buffers the kernels read, once, instead of autocast re-casting 3.9 G parameters on every step
(reference: sgm/modules/diffusionmodules/wrappers.py:87).
"""
import contextlib

import torch

BF16 = torch.bfloat16
HALF_TYPES = (torch.bfloat16, torch.float16)

# The element type the kernels run in ("compute dtype").  bf16 is the product default (libsupir_hip.so); torch.float16
# selects libsupir_hip_f16.so -- the same kernels built with fp16 MFMA operands (csrc/common.h, SUPIR_F16) -- for callers that ask
# for the reference's default diff_dtype (options/SUPIR_v0.yaml:5, test.py:67-68: model.model.dtype = torch.float16); torch.float32
# selects the fp32 service (libsupir_hip_f32.so) for `--diff_dtype fp32` / `--ae_dtype fp32` requests.  The value
# is a per-call scope set by the module that owns the request (ControlWrapper / the VAE entry points), never a process global
# that outlives a call: activations created inside the scope, the derived weight layouts (Prep caches key on it) and the library
# an op dispatches to (by the dtype of its operands) all follow it.
_CDT = [BF16]


def cdt():
    """The compute dtype of the innermost active scope (bf16 outside any scope)."""
    return _CDT[-1]


# fp32 requests (`--diff_dtype fp32` / `--ae_dtype fp32`; the reference's constructor defaults) run on the fp32 service
# (libsupir_hip_f32.so, ops_f32.py: exact-fp32 MFMA, the reference's own arithmetic for that request) when True; SUPIR_FP32_NATIVE=0
# serves them by bf16 instead -- never silently (note_downgrade).
FP32_NATIVE = __import__("os").environ.get("SUPIR_FP32_NATIVE", "1") == "1"


def as_compute_dtype(dtype):
    """Map a requested module dtype to the kernel element type: fp16 stays fp16, fp32 stays fp32 (FP32_NATIVE), everything else is bf16.
    With FP32_NATIVE off an fp32 request is a DOWNGRADE (the reference computes such a request in fp32, see note_downgrade): callers
    that own a request say so through note_downgrade before opening the scope."""
    if dtype == torch.float32 and FP32_NATIVE:
        return torch.float32
    return torch.float16 if dtype == torch.float16 else BF16


def note_downgrade(what, requested, served, why, stacklevel=3):
    """The caller asked for `requested` arithmetic and gets `served` (fewer mantissa bits): never silently.  A RuntimeWarning,
    or -- with SUPIR_STRICT_DTYPE=1 in the environment -- a RuntimeError, so that a pipeline that needs the reference's own
    precision for that request fails instead of drifting."""
    import os
    import warnings
    msg = (f"{what} is {requested} but the MI355X path serves it in {served} MFMA arithmetic with fp32 accumulation: {why}  "
           f"Set the attribute to {served} to acknowledge, or SUPIR_STRICT_DTYPE=1 to make this an error.")
    if os.environ.get("SUPIR_STRICT_DTYPE") == "1":
        raise RuntimeError(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=stacklevel)


@contextlib.contextmanager
def compute_dtype(dtype):
    _CDT.append(as_compute_dtype(dtype))
    try:
        yield
    finally:
        _CDT.pop()


def linear_w(w):
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> bf16 [N, K]."""
    return w.detach().reshape(w.shape[0], -1).to(cdt()).contiguous()


def conv3x3_w(w):
    """nn.Conv2d weight [Cout, Cin, 3, 3] -> bf16 [Cout, 3, 3, Cin] (K = (ky, kx, cin), cin fastest)."""
    return w.detach().permute(0, 2, 3, 1).to(cdt()).contiguous()


def conv3x3_w9(w):
    """nn.Conv2d weight [Cout<=8, Cin, 3, 3] -> bf16 [9, Cout, Cin] for supir_conv3x3_smallcout."""
    co, ci = w.shape[:2]
    return w.detach().permute(2, 3, 0, 1).reshape(9, co, ci).to(cdt()).contiguous()


def f32(t):
    return None if t is None else t.detach().float().contiguous()


def interleave_geglu(w, b, block=32):
    """GEGLU.proj weight [2N, K] (first half value, second half gate; sgm/modules/attention.py:89-91) ->
    rows interleaved in blocks of `block` value rows + `block` gate rows so that one wave's fragment pair holds value and
    gate of the same `block` output columns (supir_gemm_bf16 act=GEGLU): 32 for the 32x32x16-MFMA tiles of gemm.hip, 16 for
    the 16x16x32-MFMA tile 34 of gemm16.hip."""
    n2, k = w.shape
    n = n2 // 2
    assert n % block == 0 and n2 % 128 == 0
    wi = torch.stack([w[:n].reshape(n // block, block, k), w[n:].reshape(n // block, block, k)], dim=1).reshape(n2, k).contiguous()
    bi = None
    if b is not None:
        bi = torch.stack([b[:n].reshape(n // block, block), b[n:].reshape(n // block, block)], dim=1).reshape(n2).contiguous()
    return wi, bi


def fold_layernorm(w, bias, gamma, beta):
    """LayerNorm folded into the following Linear: returns (W' = bf16(gamma (.) W) [N,K], colsum[n] = sum_k W'[n,k] (fp32,
    from the ROUNDED W' so that the mean term cancels exactly), b'[n] = bias[n] + sum_k beta[k] W[n,k]).
    LayerNorm(x).W^T + bias == rstd * (x.W'^T - mean*colsum) + b'  (supir_gemm_bf16_ln)."""
    w32 = w.detach().float().reshape(w.shape[0], -1)
    wp = (w32 * gamma.detach().float()[None, :]).to(cdt()).contiguous()
    colsum = wp.float().sum(dim=1).contiguous()
    bp = (w32 * beta.detach().float()[None, :]).sum(dim=1)   # elementwise + reduction: no BLAS call on the product path
    if bias is not None:
        bp = bp + bias.detach().float()
    return wp, colsum, bp.contiguous()
