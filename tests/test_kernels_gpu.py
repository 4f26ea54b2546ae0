"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 reference of the same op, on the GPU.

Inputs are rounded to bf16 first so both sides see identical operands; the tolerance then only has to absorb
fp32-accumulation order and the final bf16 rounding of the output (rel 2^-8 per element).
Tolerances (stated): GEMM / conv / attention rel-L2 <= 4e-3, GroupNorm / LayerNorm <= 4e-3
(bf16 output rounding dominates), max-abs <= 2^-7 * max|ref| + small.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from supir_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def check(out, ref, rel=4e-3, name=""):
    out = out.float()
    ref = ref.float()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    err = (out - ref).norm() / (ref.norm() + 1e-12)
    mx = (out - ref).abs().max().item()
    bound = 2.0 ** -7 * ref.abs().max().item() + 1e-3
    assert err.item() <= rel, f"{name}: rel-L2 {err.item():.3e} > {rel} (max-abs {mx:.3e})"
    assert mx <= bound * 1.5, f"{name}: max-abs {mx:.3e} > {bound * 1.5:.3e}"


GEMM_SHAPES = [
    # (M, N, K) -- production shapes of one CFG-doubled 1024^2 step plus ragged / tiny cases
    (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (154, 1280, 2048), (8192, 640, 640),
    (8192, 5120, 640), (8192, 640, 2560), (32768, 320, 320), (2, 1280, 320), (2, 1280, 2816), (32, 320, 320),
    (77, 640, 2048), (200, 64, 64), (130, 132, 128),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_gemm_plain(M, N, K, tile):
    if tile >= 0 and M * N > 2048 * 1280:
        pytest.skip("forced tiles only on small/medium shapes")
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    out = ops.gemm(a, w, bias, tile=tile)
    ref = a.float() @ w.float().T + bias
    check(out, ref, name=f"gemm{(M, N, K)} tile{tile}")


def test_gemm_epilogues():
    B, T, N, K = 2, 96, 640, 320
    M = B * T
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    res = rnd(M, N, seed=3).to(BF)
    rb = rnd(B, N, seed=4).to(BF)
    base = a.float() @ w.float().T + bias
    # residual + alpha
    out = ops.gemm(a, w, bias, residual=res, alpha=0.5)
    check(out, 0.5 * base + res.float(), name="res+alpha")
    # rowbias (time-embedding style) + silu
    out = ops.gemm(a, w, bias, rowbias=rb, rows_per_batch=T, act=1)
    check(out, F.silu(base + rb.float().repeat_interleave(T, 0)), name="rowbias+silu")
    # fp32 output
    out = ops.gemm(a, w, None, out_dtype=torch.float32)
    check(out, a.float() @ w.float().T, rel=1e-4, name="fp32out")
    # strided A (column slice of a wider buffer) and strided C
    wide = rnd(M, K + 64).to(BF)
    cbuf = torch.zeros(M, N + 128, dtype=BF, device=DEV)
    ops.gemm(wide[:, 64:], w, bias, out=cbuf[:, 128:])
    check(cbuf[:, 128:], wide[:, 64:].float() @ w.float().T + bias, name="strided")
    assert cbuf[:, :128].abs().max().item() == 0.0


@pytest.mark.parametrize("M,K,N2", [(2048, 1280, 10240), (8192, 640, 5120), (96, 320, 2560)])
@pytest.mark.parametrize("tile", [-1, 0, 2, 4, 5, 6])
def test_gemm_geglu(M, K, N2, tile):
    """GEGLU epilogue (reference: sgm/modules/attention.py:89-91: first half value, second half gate, erf GELU)."""
    from supir_amd.weights import interleave_geglu
    a = rnd(M, K).to(BF)
    w = rnd(N2, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N2, seed=2)
    wi, bi = interleave_geglu(w, bias)
    out = ops.gemm(a, wi, bi, act=2, tile=tile)
    y = a.float() @ w.float().T + bias
    v, g = y.chunk(2, dim=-1)
    check(out, v * F.gelu(g), name="geglu")


@pytest.mark.parametrize("B,T,N,K", [(2, 1024, 1280, 1280), (2, 77, 640, 2048), (2, 16, 640, 640), (1, 4096, 640, 640)])
@pytest.mark.parametrize("tile", [-1, 3, 4, 5, 6])
def test_gemm_transposed(B, T, N, K, tile):
    a = rnd(B * T, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    Tp = (T + 63) // 64 * 64
    out = ops.gemm_t(a, w, None, B, T, Tp, tile=tile)
    ref = (a.float() @ w.float().T).view(B, T, N).permute(0, 2, 1)
    check(out[:, :, :T], ref, name="gemm_t")
    if Tp != T:
        assert out[:, :, T:].abs().max().item() == 0.0


CONV_CASES = [
    # B, H, W, Cin, Cout, stride, pad(top,left), upsample, out_hw
    (2, 32, 32, 1280, 1280, 1, (1, 1), False, None),
    (2, 64, 64, 640, 640, 1, (1, 1), False, None),
    (2, 128, 128, 320, 320, 1, (1, 1), False, None),
    (2, 32, 32, 1920, 1280, 1, (1, 1), False, None),
    (2, 64, 64, 320, 320, 2, (1, 1), False, None),          # UNet Downsample (openaimodel.py:196)
    (1, 64, 64, 128, 128, 2, (0, 0), False, (32, 32)),      # VAE Downsample, F.pad(0,1,0,1) (model.py:81-86)
    (2, 16, 16, 640, 640, 1, (1, 1), True, None),           # Upsample nearest 2x folded (openaimodel.py:145)
    (1, 24, 40, 256, 128, 1, (1, 1), False, None),          # non-square, VAE-like
    (2, 8, 8, 320, 128, 1, (1, 1), False, None),            # ZeroSFT mlp_shared
    (1, 5, 7, 64, 64, 1, (1, 1), False, None),              # ragged tiny
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [-1, 4, 5, 6])
def test_conv3x3(case, tile):
    B, H, W, Cin, Cout, stride, pad, up, out_hw = case
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    wk = w.permute(0, 2, 3, 1).contiguous()
    out = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile)
    xr = x.float().permute(0, 3, 1, 2)
    if up:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if pad == (0, 0):
        xr = F.pad(xr, (0, 1, 0, 1))
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=0)
    else:
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=1)
    check(out, ref.permute(0, 2, 3, 1), name=f"conv{case}")


def test_conv3x3_epilogue():
    """ResBlock fusion: conv + bias + emb broadcast add (openaimodel.py:338-355) and + skip (:356)."""
    B, H, W, Cin, Cout = 2, 16, 16, 320, 640
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    emb = rnd(B, Cout, seed=3).to(BF)
    res = rnd(B, H, W, Cout, seed=4).to(BF)
    wk = w.permute(0, 2, 3, 1).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    out = ops.conv3x3(x, wk, bias, rowbias=emb)
    check(out, ref + emb.float()[:, None, None, :], name="conv+emb")
    out = ops.conv3x3(x, wk, bias, residual=res)
    check(out, ref + res.float(), name="conv+res")
    out = ops.conv3x3(x, wk, bias, act=1)
    check(out, F.silu(ref), name="conv+silu")


ATTN_CASES = [(2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (1, 5, 64, 64),
              (2, 10, 16, 16), (1, 3, 200, 130), (2, 20, 1024, 4096)]


@pytest.mark.parametrize("B,H,Tq,Tk", ATTN_CASES)
def test_flash_attn(B, H, Tq, Tk):
    C = H * 64
    q = rnd(B, Tq, C).to(BF)
    k = rnd(B, Tk, C, seed=1).to(BF)
    v = rnd(B, Tk, C, seed=2).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, dtype=BF, device=DEV)
    vt[:, :, :Tk] = v.permute(0, 2, 1)
    out = ops.flash_attn(q, k, vt, B, H, Tq, Tk)
    qf, kf, vf = (t.float().view(B, -1, H, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3).reshape(B, Tq, C)
    check(out, ref, rel=6e-3, name=f"attn{(B, H, Tq, Tk)}")


def test_flash_attn_strided_qk_and_spike():
    """q,k as column slices of a fused [M, 2C] projection; a spiked key forces the online-softmax rescale branch."""
    B, H, T = 2, 10, 512
    C = H * 64
    qk = rnd(B, T, 2 * C).to(BF)
    qk[:, 300, C:] *= 12.0  # one huge key in the 5th KV tile
    v = rnd(B, T, C, seed=2).to(BF)
    vt = v.permute(0, 2, 1).contiguous()
    out = ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, B, H, T, T)
    qf, kf, vf = (t.float().reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in (qk[:, :, :C], qk[:, :, C:], v))
    ref = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3).reshape(B, T, C)
    check(out, ref, rel=6e-3, name="attn-spike")


@pytest.mark.parametrize("rows,T", [(64, 16384), (100, 1024), (7, 260)])
def test_softmax_rows(rows, T):
    s = rnd(rows, T, scale=3.0)
    out = ops.softmax_rows(s, 512 ** -0.5)
    check(out, torch.softmax(s * 512 ** -0.5, dim=-1), name="softmax")


GN_CASES = [(2, 128 * 128, 320), (2, 64 * 64, 640), (2, 32 * 32, 1280), (2, 32 * 32, 2560), (2, 64 * 64, 960),
            (2, 32 * 32, 1920), (1, 256 * 256, 128), (1, 50, 256), (2, 16, 512), (2, 9, 64)]


@pytest.mark.parametrize("B,HW,C", GN_CASES)
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(B, HW, C, silu):
    x = (rnd(B, HW, C) * 1.5 + 0.7).to(BF)
    gamma = rnd(C, seed=1) * 0.2 + 1.0
    beta = rnd(C, seed=2) * 0.2
    out = ops.groupnorm(x, gamma, beta, 1e-5, silu=silu)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    check(out, ref.permute(0, 2, 1), name=f"gn{(B, HW, C, silu)}")


def test_groupnorm_concat_and_sft():
    """ZeroSFT tail (SUPIR/modules/SUPIR_v0.py:102-113): GN over cat[h_ori, h], *(gamma+1)+beta, control_scale lerp."""
    B, HW, Cd, Cs = 2, 16 * 16, 640, 320
    C = Cd + Cs
    h_ori = rnd(B, HW, Cd).to(BF)
    h = rnd(B, HW, Cs, seed=1).to(BF)
    h_raw = rnd(B, HW, Cs, seed=5).to(BF)
    gm = (rnd(B, HW, C, seed=2) * 0.3).to(BF)
    bt = (rnd(B, HW, C, seed=3) * 0.3).to(BF)
    gamma = rnd(C, seed=1) * 0.2 + 1.0
    beta = rnd(C, seed=2) * 0.2
    cat = torch.cat([h_ori, h], -1).float()
    gn = F.group_norm(cat.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    out = ops.groupnorm(h_ori, gamma, beta, 1e-5, x2=h)
    check(out, gn, name="gn-cat")
    out = ops.groupnorm(h_ori, gamma, beta, 1e-5, x2=h, mod_g=gm, mod_b=bt)
    ref = gn * (gm.float() + 1) + bt.float()
    check(out, ref, name="sft")
    cs = 0.6
    out = ops.groupnorm(h_ori, gamma, beta, 1e-5, x2=h, mod_g=gm, mod_b=bt, control_scale=cs, x2raw=h_raw)
    raw = torch.cat([h_ori, h_raw], -1).float()
    check(out, ref * cs + raw * (1 - cs), name="sft-lerp")


@pytest.mark.parametrize("rows,C", [(2048, 1280), (8192, 640), (77, 320), (5, 2048), (3, 64)])
def test_layernorm(rows, C):
    x = (rnd(rows, C) * 2 + 0.3).to(BF)
    gamma = rnd(C, seed=1) * 0.2 + 1.0
    beta = rnd(C, seed=2) * 0.2
    out = ops.layernorm(x, gamma, beta, 1e-5)
    check(out, F.layer_norm(x.float(), (C,), gamma, beta, 1e-5), name="ln")


# (4 -> 320), (4 -> 512), (3 -> 128), (4 -> 128) take the exact-fp32 matrix-instruction form (conv3x3_smallcin_mfma_kernel: blocks of 16 flat
# pixel indices, so ragged maps -- pixel counts that are not multiples of 16, blocks straddling image rows and batch elements, fewer than
# 16 pixels -- are the edge cases); everything else (and tools knob 7) the VALU form.  Both accumulate in fp32 from fp32 operands.
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 4, 32, 32, 320), (1, 3, 40, 24, 128), (1, 4, 16, 16, 512), (1, 3, 37, 29, 128),
                                           (3, 4, 9, 7, 320), (1, 4, 5, 3, 128), (2, 4, 3, 3, 512), (2, 4, 128, 128, 320),
                                           (1, 5, 8, 8, 64), (1, 4, 7, 6, 64)])
def test_conv_smallcin(B, Cin, H, W, Cout):
    x = rnd(B, Cin, H, W)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1)
    bias = rnd(Cout, seed=2)
    add = rnd(B, H, W, Cout, seed=3).to(BF)
    out = ops.conv3x3_smallcin(x, w, bias)
    ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1)
    check(out, ref, name="smallcin")
    out_add = ops.conv3x3_smallcin(x, w, bias, add=add)
    check(out_add, ref + add.float(), name="smallcin+add")
    check(ops.conv3x3_smallcin(x, w, None), ref - bias, name="smallcin, no bias")
    assert torch.equal(out, ops.conv3x3_smallcin(x, w, bias))          # the wave-private stage hands over without a race
    # a wider destination (channel slice of a concat buffer) and the VALU form: same values up to the fp32 summation order
    wide = torch.zeros(B, H, W, Cout + 64, dtype=BF, device=x.device)
    ops.conv3x3_smallcin(x, w, bias, out=wide[..., 32:32 + Cout])
    assert torch.equal(wide[..., 32:32 + Cout], out) and not wide[..., :32].any() and not wide[..., 32 + Cout:].any()
    from supir_amd import _lib
    with _lib.tools_knob(7, 1):      # the tools build (libsupir_hip_tools.so): the only one with variant switches
        valu = ops.conv3x3_smallcin(x, w, bias, add=add)
    check(out_add, valu.float(), rel=3e-3, name="matrix-instruction form vs VALU form")
    assert ((out_add.float() - valu.float()).abs() > 0).float().mean().item() < 0.02   # they differ by last-bit ties only


# Cin = 128 with Cout 3 / 4 takes the register-resident kernel (conv3x3_c128_smallcout_kernel: a 16-lane group walks 64-pixel row segments
# with a rotating three-column window): ragged segment ends (W = 100, 65), rows of fewer pixels than the window (W = 2, 1), a single row,
# batch > 1, and a 1024-wide row (16 segments)
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 320, 32, 32, 4), (1, 512, 16, 24, 8), (1, 128, 64, 64, 3), (2, 128, 40, 100, 3),
                                            (1, 128, 33, 65, 4), (1, 128, 5, 2, 3), (2, 128, 1, 1, 3), (1, 128, 3, 1024, 3)])
def test_conv_smallcout(B, Cin, H, W, Cout):
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    w9 = w.permute(2, 3, 0, 1).reshape(9, Cout, Cin).contiguous()
    out = ops.conv3x3_smallcout(x, w9, bias)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1)
    check(out, ref, rel=1e-4, name="smallcout")


def test_pointwise():
    x = rnd(2, 8, 16, 16)
    w = rnd(8, 8, 1, 1, seed=1)
    b = rnd(8, seed=2)
    out = ops.pointwise_nchw(x, w, b, in_scale=1.0 / 0.13025)
    check(out, F.conv2d(x / 0.13025, w, b), rel=1e-5, name="pointwise")


def test_errors_are_loud():
    from supir_amd._lib import SupirHipError
    a = rnd(64, 100).to(BF)  # K % 64 != 0
    w = rnd(64, 100).to(BF)
    with pytest.raises((SupirHipError, AssertionError)):
        ops.gemm(a, w)
    with pytest.raises(SupirHipError):
        ops.gemm(a.cpu(), w.cpu())


def test_softmax_rows_padded():
    rows, T, Tp = 50, 7396 // 4, 1856  # T = 1849 (not a multiple of 4), padded to a multiple of 64
    s = rnd(rows, Tp, scale=3.0)
    out = ops.softmax_rows(s, 512 ** -0.5, valid=T)
    ref = torch.zeros_like(s)
    ref[:, :T] = torch.softmax(s[:, :T] * 512 ** -0.5, dim=-1)
    check(out, ref, name="softmax-padded")
    assert out[:, T:].abs().max().item() == 0.0


def test_groupnorm_stats_and_given():
    """Statistics-only pass and normalisation with externally supplied (pooled) statistics: the tiled-VAE GroupNorm
    (SUPIR/utils/tilevae.py:511-553)."""
    B, H, W, C = 2, 13, 9, 256
    x = (rnd(B, H, W, C) * 1.3 + 0.4).to(BF)
    gamma = rnd(C, seed=1) * 0.2 + 1.0
    beta = rnd(C, seed=2) * 0.2
    st = ops.groupnorm_stats(x)
    xf = x.float().reshape(B, H * W, 32, C // 32)
    check(st[..., 0], xf.sum(dim=(1, 3)), rel=1e-5, name="gn-sum")
    check(st[..., 1], (xf * xf).sum(dim=(1, 3)), rel=1e-5, name="gn-sumsq")
    mean = rnd(B, 32, seed=5) * 0.1 + 0.4
    var = rnd(B, 32, seed=6).abs() + 0.5
    given = torch.stack([mean, var], -1).contiguous()
    out = ops.groupnorm(x, gamma, beta, 1e-6, silu=True, given=given)
    r = x.float().reshape(B, H, W, 32, C // 32)
    r = (r - mean.view(B, 1, 1, 32, 1)) / (var.view(B, 1, 1, 32, 1) + 1e-6).sqrt()
    ref = F.silu(r.reshape(B, H, W, C) * gamma + beta)
    check(out, ref, name="gn-given")


def test_vae_attention_any_token_count():
    from tests.helpers import build_vae
    vae = build_vae(DEV)
    att = vae.decoder.mid.attn_1
    x = rnd(1, 512, 9, 7)  # 63 tokens
    with torch.no_grad():
        out = att(x)
        n = F.group_norm(x.to(BF).float(), 32, att.norm.weight, att.norm.bias, 1e-6)
        q, k, v = (F.conv2d(n, getattr(att, m).weight, getattr(att, m).bias) for m in "qkv")
        q, k, v = (t.reshape(1, 512, 63).permute(0, 2, 1)[:, None] for t in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v)[:, 0].permute(0, 2, 1).reshape(1, 512, 9, 7)
        ref = x.to(BF).float() + F.conv2d(o, att.proj_out.weight, att.proj_out.bias)
    check(out, ref, rel=1.5e-2, name="vae-attn-63")


@pytest.mark.parametrize("M,C,N", [(2048, 1280, 2560), (8192, 640, 640), (200, 320, 640), (77, 640, 1280)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 3, 5])
def test_gemm_layernorm_folding(M, C, N, tile):
    """producer emits row statistics, consumer folds LayerNorm(x).W^T + b (supir_gemm_bf16_ln) -- vs LayerNorm then Linear."""
    from supir_amd.weights import fold_layernorm
    a = rnd(M, C).to(BF)
    wp = rnd(C, C, scale=C ** -0.5, seed=1).to(BF)
    res = (rnd(M, C, seed=3) * 2 + 0.5).to(BF)
    x, st = ops.gemm_ln(a, wp, None, residual=res, emit_stats=True, tile=tile)
    xr = a.float() @ wp.float().T + res.float()
    check(x, xr, name="producer")
    xs = x.float()
    tot = st.buf[:, :st.slots].sum(dim=1)
    check(tot[:, 0], xs.sum(-1), rel=1e-4, name="rowsum")
    check(tot[:, 1], (xs * xs).sum(-1), rel=1e-4, name="rowsq")
    gamma, beta = rnd(C, seed=4) * 0.2 + 1.0, rnd(C, seed=5) * 0.2
    w = rnd(N, C, scale=C ** -0.5, seed=6)
    bias = rnd(N, seed=7)
    wf, cs, bf_ = fold_layernorm(w, bias, gamma, beta)
    ref = F.layer_norm(xs, (C,), gamma, beta, 1e-5) @ w.to(BF).float().T + bias
    out = ops.gemm_ln(x, wf, bf_, ln=st, colsum=cs, tile=tile)
    check(out, ref, rel=6e-3, name="ln-fold")
    B, T = (2, M // 2) if M % 2 == 0 else (1, M)
    Tp = (T + 63) // 64 * 64
    outt = ops.gemm_ln(x, wf, bf_, ln=st, colsum=cs, trans=(B, T, Tp), tile=tile)
    check(outt[:, :, :T], ref.view(B, T, N).permute(0, 2, 1), rel=6e-3, name="ln-fold-T")


@pytest.mark.parametrize("nbytes", [0, 128, 1000, 3 * 1280 * 1280 * 2])
@pytest.mark.parametrize("tile", [-1, 0, 3, 5])
def test_next_weight_prefetch_is_read_only(nbytes, tile):
    """supir_launch_hints.next_weight: the launch that carries the request returns bit-identical results and the prefetched buffer
    is untouched (whole 128-byte lines only, so 1000 bytes -> 7 lines); the request is an argument of that one launch (ABI 2: no
    library state), so the next launch carries none."""
    import ctypes
    from supir_amd import _lib
    lib = _lib.load()
    a = rnd(300, 640).to(BF)
    w = rnd(320, 640, scale=640 ** -0.5, seed=1).to(BF)
    b = rnd(320, seed=2)
    nxt = rnd(3 * 1280, 1280, seed=3).to(BF)
    nxt_copy = nxt.clone()
    ref = ops.gemm(a, w, b, tile=tile)
    out = torch.empty_like(ref)
    hints = _lib.LaunchHints(next_weight=nxt.data_ptr(), next_weight_bytes=nbytes, gn_partials_out=None)
    rc = lib.supir_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), out.data_ptr(), 300, 320, 640, 640, 320, b.data_ptr(), None, 0, 0, None, 0, 0, 0,
                                1.0, tile, ctypes.byref(hints), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    out2 = ops.gemm(a, w, b, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(out2, ref)
    assert torch.equal(nxt, nxt_copy)
    bad = _lib.LaunchHints(next_weight=None, next_weight_bytes=128, gn_partials_out=None)       # bytes without a pointer
    assert lib.supir_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), out.data_ptr(), 300, 320, 640, 640, 320, b.data_ptr(), None, 0, 0, None, 0, 0,
                                  0, 1.0, tile, ctypes.byref(bad), torch.cuda.current_stream().cuda_stream) == -1


def _wavelet_ref(img, levels=5):
    """fp32 torch restatement of SUPIR/utils/colorfix.py:73-107 (replicate pad + depthwise dilated 3x3 blur)."""
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=torch.float32)
    k = k[None, None].repeat(3, 1, 1, 1)
    img = img.double().cpu()
    k = k.double()
    high = torch.zeros_like(img)
    low = img
    for i in range(levels):
        r = 2 ** i
        low = F.conv2d(F.pad(img, (r,) * 4, mode="replicate"), k, groups=3, dilation=r)
        high = high + (img - low)
        img = low
    return high, low


@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (2, 3, 40, 72), (1, 3, 7, 5), (1, 3, 257, 300), (1, 3, 1, 1)])
def test_wavelet_decomposition(shape):
    """supir_wavelet_level x5 vs an fp64 torch restatement of the reference's conv-based decomposition (fp32 tolerance 1e-5;
    images smaller than the largest radius exercise the clamped == replicate-padded taps)."""
    x = rnd(*shape, seed=11)
    high, low = ops.wavelet_decomposition(x)
    rh, rl = _wavelet_ref(x)
    assert torch.isfinite(high).all() and torch.isfinite(low).all()
    for out, ref in ((high, rh), (low, rl)):
        ref = ref.to(out.device)
        err = (out.double() - ref).abs().max().item()
        assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err
    # reconstruction identity of the decomposition itself: high + low == img
    assert (high + low - x).abs().max().item() <= 1e-5
    # low band only (the style image of wavelet_reconstruction): same low, no high buffer
    h2, l2 = ops.wavelet_decomposition(x, want_high=False)
    assert h2 is None and torch.equal(l2, low)


def test_wavelet_reconstruction_vs_reference_golden():
    from supir_amd.utils.colorfix import wavelet_reconstruction
    from tests.helpers import golden, synth_tensor
    g = golden()
    out = wavelet_reconstruction(synth_tensor("wa", (1, 3, 64, 64)).to(DEV), synth_tensor("wb", (1, 3, 64, 64)).to(DEV))
    ref = g["wavelet"].to(DEV)
    assert ((out - ref).norm() / ref.norm()).item() <= 1e-5


def test_wavelet_level_rejects_aliased_buffers():
    from supir_amd import _lib
    lib = _lib.load()
    a = rnd(1, 3, 8, 8)
    b = torch.empty_like(a)
    assert lib.supir_wavelet_level(a.data_ptr(), a.data_ptr(), b.data_ptr(), 3, 8, 8, 1, 1, None) != 0
    assert lib.supir_wavelet_level(a.data_ptr(), b.data_ptr(), b.data_ptr(), 3, 8, 8, 1, 1, None) != 0
    assert lib.supir_wavelet_level(a.data_ptr(), b.data_ptr(), None, 3, 8, 8, 1, 1, None) == 0   # high = NULL: low band only
    assert lib.supir_wavelet_level(a.data_ptr(), None, None, 3, 8, 8, 1, 1, None) != 0


@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (2048, 1280, 5120), (300, 320, 640), (8192, 640, 640), (130, 136, 128)])
def test_gemm_split_k_groups(M, N, K):
    """Tile 7 of gemm.hip (128 x 128, two K groups of four waves; reachable by explicit request only -- the gemm16 family superseded
    it): validated on hardware at the start of round 2 (tools/r02_first_call.sh), kept under test as long as the code is in the library."""
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    b = rnd(N, seed=2)
    res = rnd(M, N, seed=3).to(BF)
    ref = a.float() @ w.float().t() + b
    check(ops.gemm(a, w, b, tile=7), ref, name="tile7")
    check(ops.gemm(a, w, b, residual=res, act=1, alpha=0.5, tile=7), F.silu(ref) * 0.5 + res.float(), name="tile7 epilogue")
    assert torch.equal(ops.gemm(a, w, b, tile=7), ops.gemm(a, w, b, tile=7))
    out32 = ops.gemm(a, w, b, out_dtype=torch.float32, tile=7)
    check(out32, ref, name="tile7 fp32")
    if M % 2 == 0:   # transposed (V-projection) epilogue: [B, N, Tpad]
        B, T = 2, M // 2
        Tp = (T + 63) // 64 * 64
        vt = ops.gemm_t(a.view(B, T, K), w, None, B, T, Tp, tile=7)
        ref_t = (a.float() @ w.float().t()).view(B, T, N).transpose(1, 2)
        check(vt[:, :, :T], ref_t, name="tile7 transposed")
        assert (vt[:, :, T:] == 0).all()


# ------------------------------------------------------------------------------------------ tiles 32 / 33 (csrc/gemm16.hip)
G16_SHAPES = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (8192, 640, 640), (128, 80, 128), (256, 160, 384),
              (384, 320, 256)]


_G16_TILE = {32: (128, 80), 33: (128, 160), 34: (256, 160), 35: (128, 80), 38: (128, 80)}   # 38: four waves, one K group (round 4)


@pytest.mark.parametrize("M,N,K", G16_SHAPES)
@pytest.mark.parametrize("tile", [32, 33, 34, 35, 38])
def test_gemm16_plain_and_epilogues(M, N, K, tile):
    """128 x 80 / 128 x 160 / 256 x 160 tiles (v_mfma_f32_16x16x32_bf16; two K groups per workgroup or eight waves on a 3-deep
    ring): plain, residual + alpha, row bias + SiLU, strided operands, run-to-run bitwise equality."""
    bm, bn = _G16_TILE[tile]
    if N % bn or M % bm or (tile == 35 and K < 256) or (tile == 38 and K < 128):
        pytest.skip("tile needs M % BM == 0, N % BN == 0 and at least ring-depth - 1 K steps per K group")
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    base = a.float() @ w.float().T + bias
    out = ops.gemm(a, w, bias, tile=tile)
    check(out, base, name=f"gemm16{(M, N, K)} tile{tile}")
    assert torch.equal(out, ops.gemm(a, w, bias, tile=tile))
    check(ops.gemm(a, w, None, tile=tile), a.float() @ w.float().T, name="no bias")
    res = rnd(M, N, seed=3).to(BF)
    check(ops.gemm(a, w, bias, residual=res, alpha=0.5, tile=tile), 0.5 * base + res.float(), name="res+alpha")
    Bt = 2
    rb = rnd(Bt, N, seed=4).to(BF)
    out = ops.gemm(a, w, bias, rowbias=rb, rows_per_batch=M // Bt, act=1, tile=tile)
    check(out, F.silu(base + rb.float().repeat_interleave(M // Bt, 0)), name="rowbias+silu")
    if M <= 2048:
        wide = rnd(M, K + 64).to(BF)
        cbuf = torch.zeros(M, N + 128, dtype=BF, device=DEV)
        ops.gemm(wide[:, 64:], w, bias, out=cbuf[:, 128:], tile=tile)
        check(cbuf[:, 128:], wide[:, 64:].float() @ w.float().T + bias, name="strided")
        assert cbuf[:, :128].abs().max().item() == 0.0
        # in-place residual accumulate (the transformer's x += f(x) form)
        acc = res.clone()
        ops.gemm(a, w, bias, residual=acc, out=acc, tile=tile)
        check(acc, base + res.float(), name="in-place residual")


@pytest.mark.parametrize("B,T,N,K", [(2, 1024, 1280, 1280), (2, 4096, 640, 640), (1, 256, 160, 128)])
@pytest.mark.parametrize("tile", [32, 33, 34, 35, 38])
def test_gemm16_transposed(B, T, N, K, tile):
    if tile == 35 and K < 256:
        pytest.skip("3-deep rings need two K steps per K group")
    a = rnd(B * T, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    out = ops.gemm_t(a, w, bias, B, T, T, tile=tile)
    ref = (a.float() @ w.float().T + bias).view(B, T, N).permute(0, 2, 1)
    check(out, ref, name="gemm16_t")


@pytest.mark.parametrize("M,C,N", [(2048, 1280, 2560), (8192, 640, 640), (256, 640, 1280)])
@pytest.mark.parametrize("ptile,ctile", [(32, 32), (33, 33), (32, 0), (3, 32), (0, 33), (34, 35), (35, 34), (38, 38), (38, 33)])
def test_gemm16_layernorm_folding(M, C, N, ptile, ctile):
    """Row statistics emitted by / consumed from the 16x16x32 tiles, mixed with the 32x32x16 tiles on the other side."""
    from supir_amd.weights import fold_layernorm
    a = rnd(M, C).to(BF)
    wp = rnd(C, C, scale=C ** -0.5, seed=1).to(BF)
    res = (rnd(M, C, seed=3) * 2 + 0.5).to(BF)
    x, st = ops.gemm_ln(a, wp, None, residual=res, emit_stats=True, tile=ptile)
    check(x, a.float() @ wp.float().T + res.float(), name="producer")
    xs = x.float()
    tot = st.buf[:, :st.slots].sum(dim=1)
    check(tot[:, 0], xs.sum(-1), rel=1e-4, name="rowsum")
    check(tot[:, 1], (xs * xs).sum(-1), rel=1e-4, name="rowsq")
    gamma, beta = rnd(C, seed=4) * 0.2 + 1.0, rnd(C, seed=5) * 0.2
    w = rnd(N, C, scale=C ** -0.5, seed=6)
    bias = rnd(N, seed=7)
    wf, cs, bf_ = fold_layernorm(w, bias, gamma, beta)
    ref = F.layer_norm(xs, (C,), gamma, beta, 1e-5) @ w.to(BF).float().T + bias
    out = ops.gemm_ln(x, wf, bf_, ln=st, colsum=cs, tile=ctile)
    check(out, ref, rel=6e-3, name="ln-fold")
    B, T = 2, M // 2
    outt = ops.gemm_ln(x, wf, bf_, ln=st, colsum=cs, trans=(B, T, T), tile=ctile)
    check(outt, ref.view(B, T, N).permute(0, 2, 1), rel=6e-3, name="ln-fold-T")
    fin = ops.rowstats_finalize(st, C, 1e-5)
    out2 = ops.gemm_ln(x, wf, bf_, ln=fin, colsum=cs, tile=ctile)
    check(out2, ref, rel=6e-3, name="ln-fold finalised stats")


def test_gemm16_rejects_inexact_shapes():
    from supir_amd import _lib
    a = rnd(200, 128).to(BF)
    w = rnd(80, 128, seed=1).to(BF)
    with pytest.raises(_lib.SupirHipError):
        ops.gemm(a, w, None, tile=32)          # M % 128 != 0
    a = rnd(128, 128).to(BF)
    w = rnd(96, 128, seed=1).to(BF)
    with pytest.raises(_lib.SupirHipError):
        ops.gemm(a, w, None, tile=32)          # N % 80 != 0


@pytest.mark.parametrize("M,K,N2", [(2048, 1280, 10240), (8192, 640, 5120), (256, 128, 640)])
def test_gemm16_geglu(M, K, N2):
    """GEGLU epilogue of tile 34 (value / gate interleaved per 16 rows of W), plain and with the LayerNorm fold, against the
    reference formula (sgm/modules/attention.py:89-91) and against the 32-row-interleaved tiles of gemm.hip."""
    from supir_amd.weights import fold_layernorm, interleave_geglu
    a = rnd(M, K).to(BF)
    w = rnd(N2, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N2, seed=2)
    w16, b16 = interleave_geglu(w, bias, 16)
    out = ops.gemm(a, w16, b16, act=2, tile=34)
    y = a.float() @ w.float().T + bias
    v, g = y.chunk(2, dim=-1)
    check(out, v * F.gelu(g), name="geglu16")
    assert torch.equal(out, ops.gemm(a, w16, b16, act=2, tile=34))
    w32, b32 = interleave_geglu(w, bias, 32)
    check(out, ops.gemm(a, w32, b32, act=2, tile=0).float(), rel=3e-3, name="geglu16 vs geglu32")
    # autotuned call with both layouts available picks whichever is faster and stays correct
    check(ops.gemm(a, w32, b32, act=2, alt16=(w16, b16)), v * F.gelu(g), name="geglu auto")
    # with the LayerNorm fold (the transformer's ff.net.0 as the product path calls it)
    C = K
    wp = rnd(C, C, scale=C ** -0.5, seed=5).to(BF)
    x, st = ops.gemm_ln(a, wp, None, emit_stats=True)
    gamma, beta = rnd(C, seed=6) * 0.2 + 1.0, rnd(C, seed=7) * 0.2
    wf, cs, bf_ = fold_layernorm(w.float(), bias, gamma, beta)
    wf16, bf16_ = interleave_geglu(wf, bf_, 16)
    _, cs16 = interleave_geglu(wf, cs, 16)
    yr = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.float().T + bias
    vr, gr = yr.chunk(2, dim=-1)
    check(ops.gemm_ln(x, wf16, bf16_, act=2, ln=st, colsum=cs16, tile=34), vr * F.gelu(gr), rel=8e-3, name="geglu16 ln-fold")


@pytest.mark.parametrize("M,K,N2", [(2048, 1280, 10240), (8192, 640, 5120)])
def test_geglu_erf_is_an_argument_every_tile_honours(M, K, N2, monkeypatch):
    """SUPIR_ACT_GEGLU_ERF (code 5; ops.EXACT_GELU): the reference's erf GELU (sgm/modules/attention.py:89-91) on every GEGLU-capable tile
    family -- gemm.hip (0), gemm16.hip (34), gemm_big.hip (37).  In fp32 before the output rounding the fitted form is within 2.5e-5 of erf, so
    after rounding to bf16 the two differ on a small fraction of elements by one ulp: the erf launches agree across tiles at least as well as
    the fitted ones, both meet the reference bar, and the switch does reach the kernel (the outputs are not all identical)."""
    from supir_amd.weights import interleave_geglu
    a = rnd(M, K).to(BF)
    w = rnd(N2, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N2, seed=2)
    w16, b16 = interleave_geglu(w, bias, 16)
    w32, b32 = interleave_geglu(w, bias, 32)
    y = a.float() @ w.float().T + bias
    v, g = y.chunk(2, dim=-1)
    ref = v * F.gelu(g)
    got = {}
    for exact in (False, True):
        monkeypatch.setattr(ops, "EXACT_GELU", exact)
        for t, (wq, bq) in ((0, (w32, b32)), (34, (w16, b16)), (37, (w16, b16))):
            got[exact, t] = ops.gemm(a, wq, bq, act=2, tile=t)
            check(got[exact, t], ref, name=f"geglu tile {t} exact={exact}")
    monkeypatch.setattr(ops, "EXACT_GELU", False)
    for t in (0, 34, 37):
        diff = (got[True, t].float() - got[False, t].float()).abs() > 0
        assert 0 < diff.float().mean().item() < 0.15, (t, diff.float().mean().item())      # reaches the kernel; one-ulp differences only (measured: 4 %)
        d = (got[True, t].float() - got[False, t].float()).abs()
        # one bf16 ulp of the value, plus the fitted form's own 2.5e-5 x |value operand| before the rounding (it matters where the product is tiny)
        assert bool((d <= 2.0 ** -7 * got[True, t].float().abs() + 1e-3).all())
    # erf is the arithmetic of gelu_f on every tile family (another MFMA shape, another summation order: equal to rounding)
    check(got[True, 34], got[True, 37].float(), rel=3e-3, name="erf: tile 34 vs tile 37")
    check(got[True, 0], got[True, 37].float(), rel=3e-3, name="erf: gemm.hip tile 0 vs tile 37")
    e_fit = (got[False, 37].float() - ref).abs().mean().item()
    e_erf = (got[True, 37].float() - ref).abs().mean().item()
    assert e_erf <= e_fit * 1.02, (e_erf, e_fit)


@pytest.mark.parametrize("M,K,N2", [(2048, 1280, 10240), (8192, 640, 5120), (256, 128, 640), (512, 320, 1280)])
def test_gemm_big_geglu(M, K, N2):
    """Tile 37 (csrc/gemm_big.hip: 256 x 320, activation operand global -> VGPR, GEGLU epilogue): the reference formula
    (sgm/modules/attention.py:89-91), bitwise repeatability, agreement with the 256 x 160 tile, the LayerNorm fold from partial
    and from finalised row statistics, a strided A operand, and rejection of shapes the tile does not fit."""
    from supir_amd.weights import fold_layernorm, interleave_geglu
    a = rnd(M, K).to(BF)
    w = rnd(N2, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N2, seed=2)
    w16, b16 = interleave_geglu(w, bias, 16)
    out = ops.gemm(a, w16, b16, act=2, tile=37)
    y = a.float() @ w.float().T + bias
    v, g = y.chunk(2, dim=-1)
    check(out, v * F.gelu(g), name="geglu big")
    assert torch.equal(out, ops.gemm(a, w16, b16, act=2, tile=37))
    if N2 % 160 == 0 and K >= 128:
        check(out, ops.gemm(a, w16, b16, act=2, tile=34).float(), rel=3e-3, name="geglu big vs tile 34")
    v0, g0 = (a.float() @ w.float().T).chunk(2, dim=-1)
    check(ops.gemm(a, w16, None, act=2, tile=37), v0 * F.gelu(g0), name="geglu big, no bias")
    # strided A (a column slice of a wider buffer)
    wide = torch.zeros(M, K + 64, dtype=BF, device=DEV)
    wide[:, :K] = a
    assert torch.equal(ops.gemm(wide[:, :K], w16, b16, act=2, tile=37), out)
    # LayerNorm fold, statistics as the producer GEMM leaves them (partials) and finalised
    C = K
    wp = rnd(C, C, scale=C ** -0.5, seed=5).to(BF)
    x, st = ops.gemm_ln(a, wp, None, emit_stats=True)
    gamma, beta = rnd(C, seed=6) * 0.2 + 1.0, rnd(C, seed=7) * 0.2
    wf, cs, bf_ = fold_layernorm(w.float(), bias, gamma, beta)
    wf16, bf16_ = interleave_geglu(wf, bf_, 16)
    _, cs16 = interleave_geglu(wf, cs, 16)
    yr = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.float().T + bias
    vr, gr = yr.chunk(2, dim=-1)
    o_ln = ops.gemm_ln(x, wf16, bf16_, act=2, ln=st, colsum=cs16, tile=37)
    check(o_ln, vr * F.gelu(gr), rel=8e-3, name="geglu big ln-fold")
    o_fin = ops.gemm_ln(x, wf16, bf16_, act=2, ln=ops.rowstats_finalize(st, C, 1e-5), colsum=cs16, tile=37)
    check(o_fin, vr * F.gelu(gr), rel=8e-3, name="geglu big ln-fold (finalised statistics)")
    # autotuned call with both layouts available stays correct whichever tile wins
    w32, b32 = interleave_geglu(w, bias, 32)
    check(ops.gemm(a, w32, b32, act=2, alt16=(w16, b16)), v * F.gelu(g), name="geglu auto")
    # shapes the tile does not fit are refused, not mis-computed
    from supir_amd._lib import SupirHipError
    with pytest.raises(SupirHipError):
        ops.gemm(a[: M - 64], w16, b16, act=2, tile=37)


def _unit_sums(y, B, rows_per_batch, bm):
    """(sum, sum of squares) per (batch, tile row of bm rows, 10-channel unit) of a bf16 [B * rows, C] tensor, in fp64."""
    C = y.shape[-1]
    yf = y.double().view(B, rows_per_batch // bm, bm, C // 10, 10)
    return torch.stack([yf.sum(dim=(2, 4)), (yf * yf).sum(dim=(2, 4))], dim=-1)


@pytest.mark.parametrize("case", [(2, 32, 32, 1280, 1280, 35), (2, 64, 64, 640, 640, 33), (2, 128, 128, 320, 320, 34), (2, 32, 32, 640, 1280, 32),
                                  (2, 32, 32, 1280, 1280, 38)])
def test_groupnorm_statistics_from_the_conv_epilogue(case):
    """supir_set_next_gn_partials: a gemm16 conv launch also leaves (sum, sum of squares) per (batch, tile row, 10-channel unit) of the
    bf16 values it stored; supir_groupnorm_nhwc_parts normalises with them in one launch (openaimodel.py:295-308: conv -> GroupNorm32
    -> SiLU).  Checked: the partials themselves, the GroupNorm against torch and against the two-launch path, repeatability."""
    B, H, W, Cin, Cout, tile = case
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias, emb = rnd(Cout, seed=2), rnd(B, Cout, seed=3).to(BF)
    y, part = ops.conv3x3(x, w, bias, rowbias=emb, tile=tile, gn_part=True)
    assert part is not None and part.C == Cout
    bm = ops._G16[tile][0]
    ref = _unit_sums(y.view(B * H * W, Cout), B, H * W, bm)
    assert part.buf.shape == ref.shape
    assert torch.allclose(part.buf.double(), ref, rtol=2e-5, atol=2e-3), (part.buf.double() - ref).abs().max()
    y2, part2 = ops.conv3x3(x, w, bias, rowbias=emb, tile=tile, gn_part=True)
    assert torch.equal(y, y2) and torch.equal(part.buf, part2.buf)
    assert torch.equal(y, ops.conv3x3(x, w, bias, rowbias=emb, tile=tile))          # the request does not change the output
    g, b = rnd(Cout, seed=4) * 0.2 + 1.0, rnd(Cout, seed=5) * 0.2
    o_parts = ops.groupnorm(y, g, b, 1e-5, silu=True, part=part)
    o_two = ops.groupnorm(y, g, b, 1e-5, silu=True)
    gn = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, g, b, 1e-5)).permute(0, 2, 3, 1)
    check(o_parts, gn, name="gn from conv partials")
    check(o_parts, o_two.float(), rel=1e-3, name="gn partials vs two-launch")
    assert torch.equal(o_parts, ops.groupnorm(y, g, b, 1e-5, silu=True, part=part))
    # in place, as the ResBlock uses it
    yc = y.clone()
    ops.groupnorm(yc, g, b, 1e-5, silu=True, out=yc, part=part)
    assert torch.equal(yc, o_parts)


def test_groupnorm_statistics_from_gemm_epilogues_and_concat():
    """proj_out-style GEMM (+ residual) as the producer, and the concat GroupNorm of a decoder ResBlock (1280 + 640 channels: groups of
    60 channels straddle neither 10-channel units nor the two sources' own partial buffers, which have different chunk counts)."""
    B, T = 2, 1024
    a = rnd(B * T, 1280).to(BF)
    w = rnd(1280, 1280, scale=1280 ** -0.5, seed=1).to(BF)
    res = rnd(B * T, 1280, seed=2).to(BF)
    h, p1 = ops.gemm(a, w, rnd(1280, seed=3), residual=res, rows_per_batch=T, tile=35, gn_part=True)
    assert p1 is not None and p1.nchunk == T // 128
    assert torch.allclose(p1.buf.double(), _unit_sums(h, B, T, 128), rtol=2e-5, atol=2e-3)
    x2 = rnd(B, 32, 32, 320, seed=4).to(BF)
    w2 = rnd(640, 3, 3, 320, scale=(9 * 320) ** -0.5, seed=5).to(BF)
    skip, p2 = ops.conv3x3(x2, w2, None, tile=34, gn_part=True)                    # 640 channels, another tile -> another chunk count
    assert p2 is not None and p2.nchunk == 1024 // 256 != p1.nchunk
    g, b = rnd(1920, seed=6) * 0.2 + 1.0, rnd(1920, seed=7) * 0.2
    hv = h.view(B, 32, 32, 1280)
    o = ops.groupnorm(hv, g, b, 1e-5, silu=True, x2=skip, part=p1, part2=p2)
    cat = torch.cat([hv, skip], -1).float()
    gn = F.silu(F.group_norm(cat.permute(0, 3, 1, 2), 32, g, b, 1e-5)).permute(0, 2, 3, 1)
    check(o, gn, name="concat gn from two producers")
    check(o, ops.groupnorm(hv, g, b, 1e-5, silu=True, x2=skip).float(), rel=1e-3, name="concat gn partials vs two-launch")
    # a tile that cannot emit the statistics says so (None) and the output is still right; a missing part2 falls back to two launches
    h3, p3 = ops.gemm(a, w, None, rows_per_batch=T, tile=3, gn_part=True)
    assert p3 is None
    o2 = ops.groupnorm(hv, g, b, 1e-5, silu=True, x2=skip, part=p1)
    assert torch.equal(o2, ops.groupnorm(hv, g, b, 1e-5, silu=True, x2=skip))


G16_CONV_CASES = [
    # B, H, W, Cin, Cout, stride, pad(top,left), upsample, out_hw   (UNet / VAE shapes whose output grid is an exact tile multiple)
    (2, 32, 32, 1280, 1280, 1, (1, 1), False, None),
    (2, 32, 32, 2560, 1280, 1, (1, 1), False, None),
    (2, 64, 64, 640, 640, 1, (1, 1), False, None),
    (2, 128, 128, 320, 320, 1, (1, 1), False, None),        # Cin % 128 != 0: only the one-K-group tile 34
    (2, 128, 128, 320, 320, 2, (1, 1), False, None),        # UNet Downsample (openaimodel.py:196)
    (1, 64, 64, 128, 160, 2, (0, 0), False, (32, 32)),      # VAE-style asymmetric padding (model.py:81-86)
    (2, 16, 16, 640, 640, 1, (1, 1), True, None),           # Upsample nearest 2x folded (openaimodel.py:145)
    (2, 32, 32, 128, 2560, 1, (1, 1), False, None),         # ZeroSFT gamma|beta conv (SUPIR_v0.py:79-87), K = 1152
    (1, 16, 24, 256, 160, 1, (1, 1), False, None),          # non-square
]


@pytest.mark.parametrize("case", G16_CONV_CASES)
@pytest.mark.parametrize("tile", [32, 33, 34, 35, 38])
def test_gemm16_conv3x3(case, tile):
    """Implicit-GEMM 3x3 convolution on the 16x16x32 tiles: plain, and with bias + time-embedding row bias + SiLU + residual."""
    B, H, W, Cin, Cout, stride, pad, up, out_hw = case
    bm, bn = _G16_TILE[tile]
    ks = 1 if tile in (34, 38) else 2
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    wk = w.permute(0, 2, 3, 1).contiguous()
    xr = x.float().permute(0, 3, 1, 2)
    if up:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if out_hw is not None:
        xr = F.pad(xr, (0, 1, 0, 1))
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=0)
    else:
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=1)
    OH, OW = ref.shape[2:]
    if (B * OH * OW) % bm or Cout % bn or Cin % (64 * ks):
        pytest.skip("not an exact fit for this tile")
    ref = ref.permute(0, 2, 3, 1)
    out = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile)
    check(out, ref, name=f"conv16 {case} tile{tile}")
    assert torch.equal(out, ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile))
    rb = rnd(B, Cout, seed=4).to(BF)
    res = rnd(B, OH, OW, Cout, seed=5).to(BF)
    out = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, rowbias=rb, residual=res, act=1, alpha=0.5,
                      tile=tile)
    check(out, 0.5 * F.silu(ref + rb.float()[:, None, None, :]) + res.float(), name="conv16 epilogue")


HALO_TILE = {48: (128, 80, 2, 32), 49: (128, 160, 2, 64), 50: (256, 160, 1, 32), 51: (256, 160, 1, 64)}     # tile: BM, BN, K groups, map width
HALO_CASES = [
    # B, H, W, Cin, Cout     (stride 1, pad 1: ResBlock in_layers / out_layers, sgm/modules/diffusionmodules/openaimodel.py:260-264, 295-308)
    (2, 32, 32, 1280, 1280),      # the 17 convolutions of the 1280-wide levels
    (2, 32, 32, 2560, 1280),      # decoder ResBlock on the skip concat
    (2, 32, 32, 1920, 1280),
    (8, 32, 32, 1280, 1280),      # tile batch / num_samples = 4
    (2, 64, 64, 640, 640),
    (2, 64, 64, 1280, 640),
    (4, 64, 64, 640, 640),
    (1, 8, 32, 128, 160),         # ONE tile row block of 256 rows; two chunks (both from the prologue), top and bottom halo rows both outside the image
    (1, 4, 32, 64, 80),           # one chunk: no halo tile is ever loaded inside the loop
    (1, 16, 32, 192, 160),        # odd number of chunks (one K group form only)
    (3, 32, 32, 320, 160),        # five chunks; batch of 3: tiles of different images
    (1, 2, 64, 256, 160),         # 128 rows = 2 map rows of 64
    (2, 12, 64, 384, 320),
    (2, 32, 32, 128, 2560),       # ZeroSFT gamma | beta convolution (SUPIR/modules/SUPIR_v0.py:79-87)
]


@pytest.mark.parametrize("case", HALO_CASES)
@pytest.mark.parametrize("tile", [48, 49, 50, 51])
def test_conv3x3_halo_tiles(case, tile):
    """Tiles 48-51 of csrc/gemm16.hip (round 6): the LDS-staged HALO form -- a tile is whole rows of the map; per 64-channel chunk its
    (rows + 2) x (W + 2) input pixels are staged once (zero page outside the image) and the nine taps read token fragments at shifted,
    swizzled LDS addresses; K order (chunk, tap); two halo buffers refilled inside the loop under a counted wait.  Held against torch fp32,
    against the implicit-GEMM tile that ran these layers before (another fp32 summation order: equal to rounding), bitwise repeatable over
    several launches (a late LDS-DMA piece or an early refill would differ run to run), with the full epilogue (bias, time-embedding row bias,
    SiLU, alpha, residual) and the GroupNorm partials of the epilogue."""
    B, H, W, Cin, Cout = case
    bm, bn, ks, hw = HALO_TILE[tile]
    if W != hw or (H * W) % bm or Cout % bn or Cin % (64 * ks):
        pytest.skip("not a shape of this halo tile")
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    wk = w.permute(0, 2, 3, 1).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    out = ops.conv3x3(x, wk, bias, tile=tile)
    check(out, ref, name=f"halo conv {case} tile{tile}")
    for _ in range(4):
        assert torch.equal(out, ops.conv3x3(x, wk, bias, tile=tile))
    old = ops.conv3x3(x, wk, bias, tile=4)
    check(out, old.float(), rel=3e-3, name="halo vs implicit-GEMM tile 4")
    # a strided input (channel slice of a wider buffer) and a strided output
    wide = torch.zeros(B, H, W, Cin + 64, dtype=BF, device=DEV)
    wide[..., 32:32 + Cin] = x
    wo = torch.zeros(B, H, W, Cout + 16, dtype=BF, device=DEV)
    ops.conv3x3(wide[..., 32:32 + Cin], wk, bias, tile=tile, out=wo[..., 8:8 + Cout])
    assert torch.equal(wo[..., 8:8 + Cout], out) and not wo[..., :8].any() and not wo[..., 8 + Cout:].any()
    rb = rnd(B, Cout, seed=4).to(BF)
    res = rnd(B, H, W, Cout, seed=5).to(BF)
    o2 = ops.conv3x3(x, wk, bias, rowbias=rb, residual=res, act=1, alpha=0.5, tile=tile)
    check(o2, 0.5 * F.silu(ref + rb.float()[:, None, None, :]) + res.float(), name="halo conv epilogue")
    # GroupNorm partials from the epilogue: sums of the stored bf16 values per (batch, tile row block, 10-channel unit)
    o3, part = ops.conv3x3(x, wk, bias, tile=tile, gn_part=True)
    assert torch.equal(o3, out)
    if part is not None:
        units = o3.float().view(B, (H * W) // bm, bm, Cout // 10, 10)
        torch.testing.assert_close(part.buf[..., 0], units.sum(dim=(2, 4)), rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(part.buf[..., 1], (units * units).sum(dim=(2, 4)), rtol=2e-3, atol=2e-2)


def test_conv3x3_autotune_may_pick_a_halo_tile_and_stays_correct():
    """tile = -1: the halo tiles are candidates of the autotuner for the shapes they fit; whatever wins, the result meets the bar."""
    for (B, H, W, Cin, Cout) in ((2, 32, 32, 1280, 1280), (2, 64, 64, 640, 640)):
        x = rnd(B, H, W, Cin).to(BF)
        w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
        wk = w.permute(0, 2, 3, 1).contiguous()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1).permute(0, 2, 3, 1)
        check(ops.conv3x3(x, wk, None), ref, name="halo autotune")


G16_VAE_CONV_CASES = [
    (1, 64, 128, 128, 128, 1, (1, 1), False, None),         # 8192 rows: sixteen 512-row tiles (tile 45)
    # B, H, W, Cin, Cout, stride, pad(top,left), upsample, out_hw      (sgm/modules/diffusionmodules/model.py:55-148, 571-743)
    (1, 64, 64, 128, 128, 1, (1, 1), False, None),          # ResnetBlock at the full-resolution level (1024^2 in production)
    (1, 32, 32, 256, 256, 1, (1, 1), False, None),
    (1, 32, 32, 128, 256, 1, (1, 1), False, None),          # channel change
    (1, 16, 16, 512, 512, 1, (1, 1), True, None),           # Upsample: nearest 2x folded into the gather
    (1, 64, 64, 128, 128, 2, (0, 0), False, (32, 32)),      # Downsample: stride 2, pad (0,1,0,1)
    (2, 16, 16, 512, 256, 1, (1, 1), False, None),
]


@pytest.mark.parametrize("case", G16_VAE_CONV_CASES)
@pytest.mark.parametrize("tile", [39, 40, 42, 45])
def test_gemm16_vae_tiles_conv3x3(case, tile):
    """Tiles 39 (256 x 128) / 40 (256 x 256, one K slice of fragments in registers at a time) / 42 (256 x 256 on the eight-phase
    ping-pong schedule, round 5) of csrc/gemm16.hip on the VAE's convolution shapes: against torch fp32, against the gemm.hip tile that
    ran them before, repeatable, epilogue terms; tile 42 accumulates in tile 40's K order: BITWISE tile 40."""
    B, H, W, Cin, Cout, stride, pad, up, out_hw = case
    bn = 128 if tile in (39, 45) else 256
    bm = 512 if tile == 45 else 256          # tile 45 (round 5): 512 x 128 on the eight-phase schedule, the 128-channel layers
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    wk = w.permute(0, 2, 3, 1).contiguous()
    xr = x.float().permute(0, 3, 1, 2)
    if up:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if out_hw is not None:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.float(), bias, stride=stride, padding=0)
    else:
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=1)
    OH, OW = ref.shape[2:]
    if (B * OH * OW) % bm or Cout % bn or (tile in (42, 45) and (OH * OW) % bm):
        pytest.skip("not an exact fit for this tile")
    ref = ref.permute(0, 2, 3, 1)
    out = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile)
    check(out, ref, name=f"conv16 {case} tile{tile}")
    assert torch.equal(out, ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile))
    check(out, ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=4).float(), rel=3e-3, name="vs gemm.hip tile 4")
    if tile in (42, 45):
        # tile 42's / 45's convolutions run their K loop as (channel chunk, tap) -- the nine taps of a 64-channel chunk back to back, so that a
        # tile's three input rows stay in L2 -- where every other tile runs (tap, chunk): another fp32 summation order, equal to rounding.
        # With the tap-major order forced (tools knob 6) it accumulates exactly as tile 40: BITWISE, on every repetition (a racy hand-off
        # -- LDS-DMA landing late, a half-tile restaged early -- shows up as run-to-run differences)
        from supir_amd import _lib
        o40 = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=40 if tile == 42 else 39)
        check(out, o40.float(), rel=2e-3, name="chunk-major vs tap-major K order")
        with _lib.tools_knob(6, 1):
            for _ in range(3):
                assert torch.equal(ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile), o40)
        for _ in range(3):
            assert torch.equal(ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, tile=tile), out)
    res = rnd(B, OH, OW, Cout, seed=5).to(BF)
    out = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, residual=res, act=1, alpha=0.5, tile=tile)
    check(out, 0.5 * F.silu(ref) + res.float(), name="conv16 epilogue")
    # GroupNorm statistics from the epilogue, in 4-channel units (the VAE's groups are 4 / 8 / 16 channels wide): the partials
    # themselves, the request leaving the output untouched, supir_groupnorm_parts_finalize -> `given` -> one apply launch against torch
    # and against the two-launch path, repeatability, in place (ResnetBlock.norm2, model.py:136-139)
    y = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, residual=res, tile=tile)
    y2, part = ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, residual=res, tile=tile, gn_part=True)
    assert part is not None and part.unit == 4 and part.C == Cout and torch.equal(y, y2)
    yf = y.double().view(B, OH * OW // bm, bm, Cout // 4, 4)
    ref_part = torch.stack([yf.sum(dim=(2, 4)), (yf * yf).sum(dim=(2, 4))], dim=-1)
    assert part.buf.shape == ref_part.shape
    assert torch.allclose(part.buf.double(), ref_part, rtol=2e-5, atol=2e-3), (part.buf.double() - ref_part).abs().max()
    g, b_ = rnd(Cout, seed=6) * 0.2 + 1.0, rnd(Cout, seed=7) * 0.2
    o_parts = ops.groupnorm(y, g, b_, 1e-6, silu=True, part=part)
    o_two = ops.groupnorm(y, g, b_, 1e-6, silu=True)
    gn = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, g, b_, 1e-6)).permute(0, 2, 3, 1)
    check(o_parts, gn, name="gn from 4-channel conv partials")
    check(o_parts, o_two.float(), rel=1e-3, name="gn partials vs two-launch")
    assert torch.equal(o_parts, ops.groupnorm(y, g, b_, 1e-6, silu=True, part=part))
    yc = y.clone()
    ops.groupnorm(yc, g, b_, 1e-6, silu=True, out=yc, part=part)
    assert torch.equal(yc, o_parts)


@pytest.mark.parametrize("M,N,K", [(4096, 512, 512), (1024, 256, 128), (512, 128, 256), (16384, 512, 512), (8192, 1280, 5120), (65536, 256, 192)])
@pytest.mark.parametrize("tile", [39, 40, 42, 45])
def test_gemm16_vae_tiles_plain(M, N, K, tile):
    """The same tiles as plain GEMMs (the VAE's 1x1 convolutions: nin_shortcut model.py:124, attention q / k / v / proj_out :164-175)."""
    bn = 128 if tile in (39, 45) else 256
    if N % bn or (tile in (39, 45) and K < 128) or (tile == 45 and M % 512):
        pytest.skip("not an exact fit for this tile")
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    base = a.float() @ w.float().T + bias
    out = ops.gemm(a, w, bias, tile=tile)
    check(out, base, name=f"gemm16{(M, N, K)} tile{tile}")
    assert torch.equal(out, ops.gemm(a, w, bias, tile=tile))
    if tile in (42, 45):       # same MFMA, same K order as tile 40 / 39: bitwise, on every repetition (a racy LDS hand-off would differ run to run)
        for _ in range(3):
            assert torch.equal(ops.gemm(a, w, bias, tile=tile), ops.gemm(a, w, bias, tile=40 if tile == 42 else 39))
    res = rnd(M, N, seed=3).to(BF)
    check(ops.gemm(a, w, bias, residual=res, alpha=0.5, tile=tile), 0.5 * base + res.float(), name="res+alpha")
    acc = res.clone()
    ops.gemm(a, w, bias, residual=acc, out=acc, tile=tile)                          # in-place residual (x += f(x))
    check(acc, base + res.float(), name="in-place residual")
    if N % 128 == 0 and M % 512 == 0:   # partials from a plain GEMM (AttnBlock.proj_out + residual, model.py:192), two batch elements
        bm_ = 512 if tile == 45 else 256
        if (M // 2) % bm_:
            return
        y, part = ops.gemm(a, w, bias, residual=res, rows_per_batch=M // 2, tile=tile, gn_part=True)
        assert part is not None and part.unit == 4 and part.nchunk == M // 2 // bm_
        yf = y.double().view(2, M // 2 // bm_, bm_, N // 4, 4)
        assert torch.allclose(part.buf.double(), torch.stack([yf.sum(dim=(2, 4)), (yf * yf).sum(dim=(2, 4))], dim=-1), rtol=2e-5, atol=2e-3)
        g, b_ = rnd(N, seed=6) * 0.2 + 1.0, rnd(N, seed=7) * 0.2
        yv = y.view(2, M // 2, N)
        check(ops.groupnorm(yv, g, b_, 1e-6, part=part), F.group_norm(yv.float().permute(0, 2, 1), 32, g, b_, 1e-6).permute(0, 2, 1),
              name="gn from 4-channel gemm partials")
    from supir_amd._lib import SupirHipError
    with pytest.raises(SupirHipError):
        ops.gemm_t(a.view(1, M, K), w, None, 1, M, M, tile=tile)                   # no transposed form
    # autotuned call: N % 80 != 0, so 39 / 40 are among the candidates; whichever tile wins the result is right
    check(ops.gemm(a, w, bias), base, name="auto")


@pytest.mark.parametrize("B,T,C", [(2, 1024, 1280), (2, 4096, 640), (1, 256, 320)])
def test_gemm_qkv_fused(B, T, C):
    """supir_gemm_bf16_qkv: q | k written normally, v transposed per batch, one launch; with and without the LayerNorm fold,
    against the two separate projections and against fp32."""
    from supir_amd.weights import fold_layernorm
    M, inner = B * T, C
    if not ops.gemm_qkv_supported(M, 3 * inner, 2 * inner, C, T):
        pytest.skip("shape not supported by the fused kernel")
    a = rnd(M, C).to(BF)
    w = rnd(3 * inner, C, scale=C ** -0.5, seed=1).to(BF)
    ref = a.float() @ w.float().T
    qk, vt = ops.gemm_qkv(a, w, None, B, T, 2 * inner)
    check(qk.view(M, 2 * inner), ref[:, :2 * inner], name="qkv: q|k")
    check(vt, ref[:, 2 * inner:].view(B, T, inner).permute(0, 2, 1), name="qkv: v^T")
    qk2, vt2 = ops.gemm_qkv(a, w, None, B, T, 2 * inner)
    assert torch.equal(qk, qk2) and torch.equal(vt, vt2)
    # LayerNorm fold
    wp = rnd(C, C, scale=C ** -0.5, seed=5).to(BF)
    x, st = ops.gemm_ln(a, wp, None, emit_stats=True)
    gamma, beta = rnd(C, seed=6) * 0.2 + 1.0, rnd(C, seed=7) * 0.2
    wf, cs, bf_ = fold_layernorm(w.float(), None, gamma, beta)
    refn = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.float().T
    qk, vt = ops.gemm_qkv(x, wf, bf_, B, T, 2 * inner, ln=st, colsum=cs)
    check(qk.view(M, 2 * inner), refn[:, :2 * inner], rel=6e-3, name="qkv ln: q|k")
    check(vt, refn[:, 2 * inner:].view(B, T, inner).permute(0, 2, 1), rel=6e-3, name="qkv ln: v^T")
    sep = ops.gemm_ln(x, wf[:2 * inner].contiguous(), bf_[:2 * inner].contiguous(), ln=st, colsum=cs[:2 * inner].contiguous())
    check(qk.view(M, 2 * inner), sep.float(), rel=3e-3, name="qkv vs separate")
    # the two tile widths of the fused launch (256 x 160 / 256 x 128; the launch takes the one with fewer workgroup rounds x columns)
    # accumulate every output element over K in the same order: forced either way (tools-only knob 4) the results are bitwise equal
    if (3 * inner) % 160 == 0 and (3 * inner) % 128 == 0:
        from supir_amd import _lib
        got = []
        for width in (1, 2):
            with _lib.tools_knob(4, width):
                got.append(ops.gemm_qkv(x, wf, bf_, B, T, 2 * inner, ln=st, colsum=cs))
        for g in got[1:]:
            assert torch.equal(got[0][0], g[0]) and torch.equal(got[0][1], g[1])
        assert torch.equal(got[0][0], qk) and torch.equal(got[0][1], vt)


@pytest.mark.parametrize("tile", [-1, 3, 32])
def test_gemm_gelu_and_quick_gelu_epilogues(tile):
    """act 3 (erf GELU: OpenCLIP text tower MLP) and act 4 (QuickGELU x * sigmoid(1.702 x): OpenAI CLIP text tower MLP)."""
    M, N, K = (256, 320, 256) if tile == 32 else (154, 3072, 768)
    a = rnd(M, K).to(BF)
    w = rnd(N, K, scale=K ** -0.5, seed=1).to(BF)
    bias = rnd(N, seed=2)
    y = a.float() @ w.float().T + bias
    check(ops.gemm(a, w, bias, act=3, tile=tile), F.gelu(y), name="gelu")
    check(ops.gemm(a, w, bias, act=4, tile=tile), y * torch.sigmoid(1.702 * y), name="quick_gelu")


@pytest.mark.parametrize("B,H,T", [(2, 12, 77), (1, 20, 77), (2, 4, 200), (1, 2, 1024)])
def test_flash_attn_causal(B, H, T):
    """supir_flash_attn_d64_ex flags bit 0: causal mask (text towers), against SDPA(is_causal=True)."""
    C = H * 64
    q, k, v = rnd(B, T, C).to(BF), rnd(B, T, C, seed=1).to(BF), rnd(B, T, C, seed=2).to(BF)
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, dtype=BF, device=DEV)
    vt[:, :, :T] = v.permute(0, 2, 1)
    out = ops.flash_attn(q, k, vt, B, H, T, T, causal=True)
    sp = lambda t: t.float().view(B, T, H, 64).permute(0, 2, 1, 3)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), is_causal=True).permute(0, 2, 1, 3).reshape(B, T, C)
    check(out, ref, rel=6e-3, name="causal attention")
    # the first query only sees the first key: its output row is exactly v[0]
    assert torch.equal(out[:, 0], v[:, 0])


@pytest.mark.parametrize("B,H,W,Cin,Cout,act,base", [(2, 32, 32, 1280, 128, 1, 3), (2, 32, 32, 640, 128, 1, 1), (2, 64, 64, 640, 128, 1, 2),
                                                    (1, 24, 20, 128, 64, 0, 3), (2, 16, 16, 320, 132, 1, 3)])
def test_conv3x3_tap_split(B, H, W, Cin, Cout, act, base):
    """supir_conv3x3_bf16_splitk + supir_splitk_finalize (tile codes 64 + t of ops.conv3x3): nine per-tap fp32 partials summed in a
    fixed order, bias + SiLU in the finalize pass -- against fp32 conv2d and against the single-launch form (same inputs; only the fp32
    summation order over taps differs), bitwise repeatable."""
    x = rnd(B, H, W, Cin).to(BF)
    w = rnd(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5, seed=1).to(BF)
    bias = rnd(Cout, seed=2)
    out = ops.conv3x3(x, w, bias, act=act, tile=64 + base)
    out2 = ops.conv3x3(x, w, bias, act=act, tile=64 + base)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
    check(out, ref, name=f"conv split9 {(B, H, W, Cin, Cout)}")
    assert torch.equal(out, out2)
    one = ops.conv3x3(x, w, bias, act=act, tile=3)
    assert ((out.float() - one.float()).norm() / one.float().norm()).item() <= 3e-3
    with pytest.raises(Exception):
        ops.conv3x3(x, w, bias, act=act, residual=out, tile=64 + base)      # no residual epilogue in the split form


XATTN_CASES = [(2, 20, 1024, 77, 1280), (2, 10, 4096, 77, 640), (1, 5, 128, 64, 320), (1, 3, 256, 128, 192), (2, 2, 128, 1, 256),
               (1, 4, 384, 33, 256), (1, 2, 128, 65, 192)]


def _xattn_operands(B, H, T, Tk, C):
    N = H * 64
    x = (rnd(B, T, C) * 1.5 + 0.3).to(BF)
    k = rnd(B, Tk, N, seed=1).to(BF)
    k[:, Tk // 2] *= 3.0      # one spiked key: the second half tile has to rescale what the first accumulated (or the other way round)
    v = rnd(B, Tk, N, seed=2).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, N, Tp, dtype=BF, device=DEV)
    vt[:, :, :Tk] = v.permute(0, 2, 1)
    w = rnd(N, C, scale=C ** -0.5, seed=6)
    return x, k, v, vt, w


def _attn_ref(q, k, v, B, H):
    qf, kf, vf = (t.float().reshape(B, -1, H, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    return F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3).reshape(B, -1, H * 64)


@pytest.mark.parametrize("B,H,T,Tk,C", XATTN_CASES)
def test_xattn_q_with_layernorm_fold(B, H, T, Tk, C):
    """supir_xattn_q_d64 (to_q with the LayerNorm fold + text cross-attention in one launch) vs LayerNorm -> Linear -> attention in
    fp32, and vs the two-launch HIP path it replaces (same operands; they differ by q's single bf16 rounding only)."""
    from supir_amd.weights import fold_layernorm
    N = H * 64
    x, k, v, vt, w = _xattn_operands(B, H, T, Tk, C)
    assert ops.xattn_q_supported(B, T, C, H, Tk)
    # the row statistics as a producer GEMM leaves them (identity-free: a real producer launch)
    wp = rnd(C, C, scale=C ** -0.5, seed=9).to(BF)
    xs, st = ops.gemm_ln(x.view(B * T, C), wp, None, residual=x.view(B * T, C), emit_stats=True)
    xs = xs.view(B, T, C)
    gamma, beta = rnd(C, seed=4) * 0.2 + 1.0, rnd(C, seed=5) * 0.2
    wf, cs, bf_ = fold_layernorm(w, None, gamma, beta)
    out = ops.xattn_q(xs, wf, bf_, k, vt, B, H, T, Tk, ln=st, colsum=cs)
    q_ref = F.layer_norm(xs.float(), (C,), gamma, beta, 1e-5) @ w.to(BF).float().T
    ref = _attn_ref(q_ref, k, v, B, H)
    check(out, ref, rel=8e-3, name="xattn-ln")
    two = ops.flash_attn(ops.gemm_ln(xs, wf, bf_, ln=st, colsum=cs), k, vt, B, H, T, Tk)
    check(out, two.float(), rel=8e-3, name="xattn-vs-two-launches")
    # finalised statistics (slots == 0) take the other branch of the kernel
    out_f = ops.xattn_q(xs, wf, bf_, k, vt, B, H, T, Tk, ln=ops.rowstats_finalize(st, C, 1e-5), colsum=cs)
    check(out_f, out.float(), rel=2e-3, name="xattn-finalised-stats")


@pytest.mark.parametrize("B,H,T,Tk,C", [(2, 20, 1024, 77, 1280), (1, 3, 256, 128, 192)])
def test_xattn_q_plain_strided_and_prefetch(B, H, T, Tk, C):
    """No LayerNorm / bias; x, k and the output as column slices of wider buffers; a next-weight request changes nothing."""
    from supir_amd import _lib
    N = H * 64
    x, k, v, vt, w = _xattn_operands(B, H, T, Tk, C)
    wb = w.to(BF)
    xw = torch.zeros(B, T, C + 64, dtype=BF, device=DEV)
    xw[:, :, :C] = x
    kw = torch.zeros(B, Tk, 2 * N, dtype=BF, device=DEV)
    kw[:, :, N:] = k
    ow = torch.full((B, T, N + 128), 7.0, dtype=BF, device=DEV)
    out = ops.xattn_q(xw[:, :, :C], wb, None, kw[:, :, N:], vt, B, H, T, Tk, out=ow[:, :, 64:64 + N])
    ref = _attn_ref(x.float() @ wb.float().T, k, v, B, H)
    check(out, ref, rel=8e-3, name="xattn-plain")
    assert (ow[:, :, :64] == 7.0).all() and (ow[:, :, 64 + N:] == 7.0).all()
    lib = _lib.load()
    nxt = rnd(1280, 1280, seed=3).to(BF)
    nxt_copy = nxt.clone()
    hints = _lib.LaunchHints(next_weight=nxt.data_ptr(), next_weight_bytes=nxt.numel() * 2, gn_partials_out=None)
    import ctypes
    o2 = torch.empty(B, T, N, dtype=BF, device=DEV)
    rc = lib.supir_xattn_q_d64(x.data_ptr(), wb.data_ptr(), None, k.data_ptr(), vt.data_ptr(), o2.data_ptr(), B, H, T, Tk, C, C, N,
                               vt.shape[-1], N, None, 0, 0, None, 1e-5, 0.125, ctypes.byref(hints), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o2, out.contiguous()) and torch.equal(nxt, nxt_copy)
    # the 2-D XCD grid (round 5: token-block chunks x head chunks per XCD) only re-orders workgroups: bitwise the 1-D order (tools knob 5 = 1)
    with _lib.tools_knob(5, 1):
        o1d = ops.xattn_q(xw[:, :, :C], wb, None, kw[:, :, N:], vt, B, H, T, Tk)
    assert torch.equal(o1d, ops.xattn_q(xw[:, :, :C], wb, None, kw[:, :, N:], vt, B, H, T, Tk))


def test_xattn_q_rejects_what_it_does_not_cover():
    from supir_amd import _lib
    lib = _lib.load()
    B, H, T, Tk, C = 1, 2, 128, 77, 256
    x, k, v, vt, w = _xattn_operands(B, H, T, Tk, C)
    wb = w.to(BF)
    o = torch.empty(B, T, H * 64, dtype=BF, device=DEV)

    def call(T_=T, Tk_=Tk, C_=C, ldx=C, x_=x):
        return lib.supir_xattn_q_d64(x_.data_ptr(), wb.data_ptr(), None, k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, T_, Tk_, C_, ldx,
                                     H * 64, vt.shape[-1], H * 64, None, 0, 0, None, 1e-5, 0.125, None, None)
    assert call() == 0
    assert call(T_=96) == -2          # not whole 128-token row blocks
    assert call(Tk_=129) == -2        # more than two key tiles
    assert call(C_=128) == -2         # fewer K steps than the ring is deep
    assert call(ldx=C + 4) == -2
    assert lib.supir_xattn_q_d64(None, wb.data_ptr(), None, k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, T, Tk, C, C, H * 64,
                                 vt.shape[-1], H * 64, None, 0, 0, None, 1e-5, 0.125, None, None) == -1
    torch.cuda.synchronize()
