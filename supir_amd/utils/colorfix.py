"""Colour fix tail step (SUPIR/utils/colorfix.py:59-119) on fp32 [N,3,H,W].

The wavelet decomposition runs on the HIP kernel `supir_wavelet_level` (5 launches per image, HBM-trivial); the reference's
F.conv2d(F.pad(..., 'replicate'), groups=3, dilation=r) went through MIOpen's naive fp32 convolution at ~7 ms per level for a
1024^2 image.  AdaIN is two reductions and an affine map and stays in torch elementwise / reduction ops."""
from .. import ops


def wavelet_decomposition(img, levels=5, want_high=True):
    return ops.wavelet_decomposition(img.float(), levels, want_high=want_high)


def wavelet_reconstruction(content, style):
    """high frequencies of `content` + low frequencies of `style` (colorfix.py:109-119); the style image's high band is never
    formed (the reference computes and discards it)."""
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style, want_high=False)[1]


def adaptive_instance_normalization(content, style, eps=1e-5):
    def ms(t):
        n, c = t.shape[:2]
        var = t.reshape(n, c, -1).var(dim=2) + eps
        return t.reshape(n, c, -1).mean(dim=2).reshape(n, c, 1, 1), var.sqrt().reshape(n, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm
