"""Colour fix tail step (SUPIR/utils/colorfix.py:59-119): HBM-trivial fp32 torch ops on [N,3,H,W]."""
import torch
import torch.nn.functional as F


def _blur(img, radius):
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=img.dtype,
                     device=img.device)[None, None].repeat(3, 1, 1, 1)
    return F.conv2d(F.pad(img, (radius,) * 4, mode="replicate"), k, groups=3, dilation=radius)


def wavelet_decomposition(img, levels=5):
    high = torch.zeros_like(img)
    for i in range(levels):
        low = _blur(img, 2 ** i)
        high = high + (img - low)
        img = low
    return high, low


def wavelet_reconstruction(content, style):
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style)[1]


def adaptive_instance_normalization(content, style, eps=1e-5):
    def ms(t):
        n, c = t.shape[:2]
        var = t.reshape(n, c, -1).var(dim=2) + eps
        return t.reshape(n, c, -1).mean(dim=2).reshape(n, c, 1, 1), var.sqrt().reshape(n, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm
