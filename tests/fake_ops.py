"""TEST-ONLY stand-in for the few `supir_amd.ops` entry points the tiled VAE calls, in plain fp32 torch, so that the HOST logic of
supir_amd/utils/tilevae.py (tile split, shape-group stacking, pooled statistics, cross-rank exchange, assembly) can run on CPU --
under gloo with world size 2 -- and be compared with the oracle's tiled forward.  Same signatures and layouts as the real ops
(channels-last activations, the derived weight layouts of supir_amd/weights.py); never imported by the product."""
import torch
import torch.nn.functional as F


def _nchw(x):
    return x.float().permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def conv3x3_smallcin(x_nchw, w, bias, add=None, out=None, dtype=None):
    y = F.conv2d(x_nchw.float(), w.float(), bias, padding=1)
    return _nhwc(y)


def conv3x3_smallcout(x, w9, bias, out=None):
    co, ci = w9.shape[1], w9.shape[2]
    w = w9.float().reshape(3, 3, co, ci).permute(2, 3, 0, 1)
    return F.conv2d(_nchw(x), w, bias, padding=1)


def conv3x3(x, w, bias=None, *, stride=1, pad=(1, 1), upsample=False, out_hw=None, rowbias=None, residual=None, act=0, alpha=1.0,
            out=None, tile=-1, gn_part=False):
    xin = _nchw(x)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    wt = w.float().permute(0, 3, 1, 2)
    if stride == 2 and pad == (0, 0):       # the VAE downsample: asymmetric (0, 1, 0, 1) padding (model.py:98-104)
        xin = F.pad(xin, (0, 1, 0, 1))
        y = F.conv2d(xin, wt, bias, stride=2)
    else:
        y = F.conv2d(xin, wt, bias, stride=stride, padding=1)
    y = _nhwc(y)
    if out_hw is not None:
        assert tuple(y.shape[1:3]) == tuple(out_hw)
    return y + residual.float() if residual is not None else y


def gemm(a, w, bias=None, *, residual=None, **kw):
    y = a.float() @ w.float().T
    if bias is not None:
        y = y + bias
    return y + residual.float() if residual is not None else y


def groupnorm_stats(x):
    B, C = x.shape[0], x.shape[-1]
    g = x.float().reshape(B, -1, 32, C // 32)
    return torch.stack([g.sum(dim=(1, 3)), (g * g).sum(dim=(1, 3))], dim=-1)


def groupnorm(x, gamma, beta, eps, *, silu=False, given=None, **kw):
    assert given is not None
    B, C = x.shape[0], x.shape[-1]
    mean = given[..., 0].repeat_interleave(C // 32, dim=1).view(B, *([1] * (x.dim() - 2)), C)
    var = given[..., 1].repeat_interleave(C // 32, dim=1).view(B, *([1] * (x.dim() - 2)), C)
    y = (x.float() - mean) * torch.rsqrt(var + eps) * gamma + beta
    return F.silu(y) if silu else y


def attend(self, n):
    """AttnBlock.attend: single-head softmax attention over the tokens of each batch element (model.py:177-192)."""
    q = n @ self.q.weight.float().reshape(self.in_channels, -1).T + self.q.bias
    k = n @ self.k.weight.float().reshape(self.in_channels, -1).T + self.k.bias
    v = n @ self.v.weight.float().reshape(self.in_channels, -1).T + self.v.bias
    return F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]


def install(monkeypatch=None):
    """Point supir_amd.utils.tilevae at this backend (and the VAE attention at the torch one); returns an undo function."""
    import sys
    from supir_amd.modules import vae as V
    from supir_amd.utils import tilevae as T
    me = sys.modules[__name__]
    old = (T.ops, V.AttnBlock.attend, T.cdt)
    T.ops, V.AttnBlock.attend, T.cdt = me, attend, (lambda: torch.float32)

    def undo():
        T.ops, V.AttnBlock.attend, T.cdt = old
    return undo
