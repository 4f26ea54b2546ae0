"""Parameter holders with the reference's state-dict key names, plus the layout conventions of the module layer.

Boundary convention (what lets these classes stand in for the reference's modules):
  * feature maps cross module boundaries as tensors of LOGICAL shape [B, C, H, W] (the reference's NCHW) whose memory is
    channels-last bf16 -- `to_nhwc` turns them into the [B, H, W, C] view the kernels consume without moving a byte,
    `to_nchw` goes back.  An fp32 / NCHW-contiguous tensor handed in by an outside caller is converted once on entry;
  * parameters stay fp32 tensors with the reference's names and shapes (drop-in `load_state_dict`); the bf16 kernel
    layouts are derived lazily and cached, keyed on (data_ptr, version) so `load_state_dict` / `.to()` invalidate them
    (and on the compute dtype: a module driven in an fp16 scope -- weights.compute_dtype -- holds fp16 layouts instead).

Parameters are allocated uninitialised (torch.empty): the reference's random init is unusable anyway (zero_module) and
real use always loads a checkpoint; tests / bench fill them with supir_amd.synth.
"""
import torch
import torch.nn as nn

from .. import weights as Wt

BF16 = torch.bfloat16
cdt = Wt.cdt   # the 16-bit element type of the current call scope (bf16 by default, fp16 inside weights.compute_dtype(torch.float16))


def to_nhwc(x):
    """[B,C,H,W] logical (any dtype/strides) -> 16-bit (compute dtype) [B,H,W,C] contiguous view/copy."""
    if x.dtype != cdt():
        x = x.to(cdt())
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def to_nchw(x_nhwc):
    """[B,H,W,C] -> logical [B,C,H,W] view (channels-last memory)."""
    return x_nhwc.permute(0, 3, 1, 2)


def attach_gn_part(t, part):
    """Tag a module OUTPUT (the exact tensor object handed to the next module) with the GroupNorm statistics its producing epilogue
    left behind (ops.GnPart).  Valid as long as nobody writes the tensor in place -- module outputs on this path are never written
    again -- and lost (harmlessly: the consumer falls back to a statistics pass) by any copy / cast / clone."""
    if part is not None:
        t._gn_part = part
    return t


def gn_part_of(t):
    return getattr(t, "_gn_part", None)


def tokens_bf16(x):
    """Token tensor in the compute dtype (bf16 by default; the name predates the fp16 build), contiguous."""
    x = x if x.dtype == cdt() else x.to(cdt())
    return x if x.is_contiguous() else x.contiguous()


class Prep:
    """Cache of a derived weight tensor; recomputed when any source parameter changes identity or version."""

    __slots__ = ("key", "val")

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, srcs, fn):
        # the compute dtype is part of the key: the derived layouts are stored in it (bf16, or fp16 inside an fp16 scope)
        key = tuple((t.data_ptr(), t._version, t.device) for t in srcs if t is not None) + (cdt(),)
        if key != self.key:
            with torch.no_grad():
                self.val = fn()
            self.key = key
        return self.val


class _Holder(nn.Module):
    def _init_prep(self):
        object.__setattr__(self, "_pw", Prep())
        object.__setattr__(self, "_pb", Prep())

    def b32(self):
        if self.bias is None:
            return None
        return self._pb.get((self.bias,), lambda: Wt.f32(self.bias))


class Linear(_Holder):
    """Holder for nn.Linear / 1x1 nn.Conv2d parameters (weight [out, in] or [out, in, 1, 1])."""

    def __init__(self, in_features, out_features, bias=True, conv1x1=False):
        super().__init__()
        shape = (out_features, in_features, 1, 1) if conv1x1 else (out_features, in_features)
        self.weight = nn.Parameter(torch.empty(shape), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features), requires_grad=False) if bias else None
        self.in_features, self.out_features = in_features, out_features
        self._init_prep()

    def w(self):
        return self._pw.get((self.weight,), lambda: Wt.linear_w(self.weight))


class Conv3x3(_Holder):
    """Holder for a 3x3 nn.Conv2d (weight [out, in, 3, 3])."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_channels), requires_grad=False)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._init_prep()
        object.__setattr__(self, "_pw9", Prep())
        object.__setattr__(self, "_pf", Prep())

    def w(self):      # [Cout,3,3,Cin] bf16 (implicit GEMM)
        return self._pw.get((self.weight,), lambda: Wt.conv3x3_w(self.weight))

    def w9(self):     # [9,Cout,Cin] bf16 (small-Cout kernel)
        return self._pw9.get((self.weight,), lambda: Wt.conv3x3_w9(self.weight))

    def wf32(self):   # fp32 [Cout,Cin,3,3] (small-Cin kernel reads the reference layout)
        return self._pf.get((self.weight,), lambda: Wt.f32(self.weight))


class Norm(_Holder):
    """Holder for GroupNorm(32) / LayerNorm affine parameters."""

    def __init__(self, channels, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(channels), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(channels), requires_grad=False)
        self.eps = eps
        self.num_channels = channels
        self._init_prep()

    def g32(self):
        return self._pw.get((self.weight,), lambda: Wt.f32(self.weight))


def GroupNorm32(channels):
    """normalization() of sgm/modules/diffusionmodules/util.py:258-276: GroupNorm(32, C), eps 1e-5."""
    return Norm(channels, 1e-5)


def Normalize(channels):
    """Normalize() of sgm/modules/attention.py:122-125 and diffusionmodules/model.py:48-51: eps 1e-6."""
    return Norm(channels, 1e-6)


class Passthrough(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference (SiLU / Dropout slots)."""

    def forward(self, x):
        return x
