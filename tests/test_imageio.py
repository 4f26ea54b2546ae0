"""Image I/O edges: PIL2Tensor / Tensor2PIL of SUPIR/util.py:60-94.

Oracle: Pillow itself (the reference's dependency, present in this image) for the resize, torch's F.interpolate for the way back.
CPU tier: the host-built coefficient tables + the kernel's integer arithmetic (emulated in numpy) against PIL, bit for bit.
GPU tier: the HIP kernels against PIL (bit-exact uint8, bit-exact fp32 tensor) and against F.interpolate.
"""
import numpy as np
import pytest
import torch

from supir_amd.utils import imageio as IO

SIZES = [(37, 29, 64, 64), (200, 150, 128, 64), (64, 64, 192, 128), (333, 127, 320, 128), (50, 80, 50, 160), (96, 64, 96, 64)]


def _pil_resize(arr, ow, oh):
    from PIL import Image
    return np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))


def _emulate(arr, ow, oh):
    H, W, C = arr.shape
    cur = arr.astype(np.int64)
    if ow != W:
        b, k, _ = IO.pillow_bicubic_coeffs(W, ow)
        out = np.zeros((H, ow, C), dtype=np.int64)
        for xx in range(ow):
            x0, n = b[xx]
            out[:, xx, :] = np.clip(((1 << 21) + np.tensordot(cur[:, x0:x0 + n, :], k[xx, :n].astype(np.int64), axes=([1], [0]))) >> 22,
                                    0, 255)
        cur = out
    if oh != H:
        b, k, _ = IO.pillow_bicubic_coeffs(H, oh)
        out = np.zeros((oh, cur.shape[1], C), dtype=np.int64)
        for yy in range(oh):
            y0, n = b[yy]
            out[yy] = np.clip(((1 << 21) + np.tensordot(k[yy, :n].astype(np.int64), cur[y0:y0 + n], axes=([0], [0]))) >> 22, 0, 255)
        cur = out
    return cur.astype(np.uint8)


@pytest.mark.parametrize("W,H,ow,oh", SIZES)
def test_coefficient_tables_reproduce_pillow_bit_for_bit(W, H, ow, oh):
    arr = np.random.default_rng(W * 1000 + H).integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(_emulate(arr, ow, oh), _pil_resize(arr, ow, oh))


def test_target_size_arithmetic_matches_reference_formula():
    """SUPIR/util.py:65-78 on a few sizes (restated inline: round to x64, min side >= min_size, upscale first)."""
    for (w, h, up, mn) in [(512, 384, 1, 1024), (1000, 1500, 2, 1024), (333, 500, 1, 256), (2048, 1024, 1, 1024)]:
        ww, hh = w * up, h * up
        w0, h0 = round(ww), round(hh)
        if min(ww, hh) < mn:
            u = mn / min(ww, hh)
            ww, hh = ww * u, hh * u
        exp = (int(np.round(ww / 64.0)) * 64, int(np.round(hh / 64.0)) * 64, w0, h0)
        assert IO.target_size(w, h, up, mn) == exp


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,ow,oh", SIZES + [(640, 480, 1408, 1024)])
def test_resample_kernel_vs_pillow_bit_exact(W, H, ow, oh):
    arr = np.random.default_rng(W + 7 * H).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref_u8 = _pil_resize(arr, ow, oh)
    out_f, out_u = IO.resize_bicubic_u8(torch.from_numpy(arr).cuda(), ow, oh, want_u8=True)
    assert np.array_equal(out_u.cpu().numpy(), ref_u8)
    ref_f = torch.tensor(ref_u8 / 255 * 2 - 1, dtype=torch.float32).permute(2, 0, 1)       # SUPIR/util.py:81-82
    assert torch.equal(out_f.cpu(), ref_f)


@pytest.mark.gpu
def test_pil2tensor_end_to_end_vs_reference_formula():
    """PIL2Tensor(img, upsacle, min_size) == the reference's function body evaluated with PIL + numpy on the host."""
    from PIL import Image
    arr = np.random.default_rng(5).integers(0, 256, (96, 130, 3), dtype=np.uint8)
    img = Image.fromarray(arr)
    x, h0, w0 = IO.PIL2Tensor(img, upsacle=2, min_size=256)
    w, h, rw0, rh0 = IO.target_size(130, 96, 2, 256)
    ref = np.array(img.resize((w, h), Image.BICUBIC)).round().clip(0, 255).astype(np.uint8)
    ref = torch.tensor(ref / 255 * 2 - 1, dtype=torch.float32).permute(2, 0, 1)
    assert (h0, w0) == (rh0, rw0) == (192, 260) and tuple(x.shape) == (3, h, w) and x.is_cuda
    assert torch.equal(x.cpu(), ref)


def test_unpremultiply_matches_pillow_convert():
    """The un-premultiply step of the alpha path (host-checkable: no kernel involved) against Pillow's own RGBa -> RGBA / La -> LA."""
    from PIL import Image
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (64, 48, 1), dtype=np.uint8)
    a[:8] = 0
    a[8:16] = 255
    c = (rng.integers(0, 256, (64, 48, 3)) * (a.astype(np.int64) + 3) // 258).astype(np.uint8)    # mostly <= alpha, some above
    for mode_pre, mode, arr in (("RGBa", "RGBA", np.concatenate([c, a], -1)), ("La", "LA", np.concatenate([c[..., :1], a], -1))):
        ref = np.asarray(Image.fromarray(np.ascontiguousarray(arr), mode_pre).convert(mode))
        got = IO.unpremultiply_u8(torch.from_numpy(np.ascontiguousarray(arr))).numpy()
        assert np.array_equal(got, ref), mode


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["RGBA", "LA"])
def test_pil2tensor_alpha_modes_vs_reference_formula(mode):
    """Images with an alpha band: Pillow resamples them premultiplied; PIL2Tensor must reproduce that (SUPIR/util.py:79-82)."""
    from PIL import Image
    rng = np.random.default_rng(21)
    nb = 4 if mode == "RGBA" else 2
    arr = rng.integers(0, 256, (70, 90, nb), dtype=np.uint8)
    arr[:20, :, -1] = 255
    arr[20:30, :, -1] = 0
    img = Image.fromarray(arr, mode)
    x, h0, w0 = IO.PIL2Tensor(img, upsacle=1, min_size=128)
    w, h, _, _ = IO.target_size(90, 70, 1, 128)
    ref = np.array(img.resize((w, h), Image.BICUBIC)).round().clip(0, 255).astype(np.uint8)
    ref = torch.tensor(ref / 255 * 2 - 1, dtype=torch.float32).permute(2, 0, 1)
    assert tuple(x.shape) == (nb, h, w) and torch.equal(x.cpu(), ref)


def test_pil2tensor_refuses_modes_it_cannot_reproduce():
    from PIL import Image
    with pytest.raises(ValueError):
        IO.PIL2Tensor(Image.new("P", (64, 64)), device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,h0,w0", [(128, 192, 96, 130), (64, 64, 200, 150), (256, 320, 256, 320), (100, 60, 37, 211)])
def test_tensor2pil_vs_torch_interpolate(H, W, h0, w0):
    """Tensor2PIL (SUPIR/util.py:86-94): fp32 result within 2e-6 of F.interpolate(bicubic); uint8 equal except where the float sits
    within that distance of an integer boundary (<= 1 LSB, < 0.1 % of the pixels)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(H * W)
    x = (torch.rand(3, H, W, generator=g) * 2.4 - 1.2).cuda()          # a little beyond [-1, 1]: exercises the clip
    out_u, out_f = IO.bicubic_resize_f32(x, h0, w0)
    ref_f = F.interpolate(x.unsqueeze(0), size=(h0, w0), mode="bicubic").squeeze(0)
    assert (out_f - ref_f).abs().max().item() <= 2e-6
    ref_u = (ref_f.permute(1, 2, 0) * 127.5 + 127.5).cpu().numpy().clip(0, 255).astype(np.uint8)
    d = np.abs(out_u.cpu().numpy().astype(int) - ref_u.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    pil = IO.Tensor2PIL(x, h0, w0)
    assert pil.size == (w0, h0) and pil.mode == "RGB"
