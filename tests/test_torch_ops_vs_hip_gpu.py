"""Pins tests/torch_ops.py (the plain-torch stand-in the CPU tier runs the host path on) to the HIP kernels, op by op, on the GPU:
the same call with the same operands through `supir_amd.ops` (C ABI -> gfx950 kernels) and through `tests.torch_ops`, compared at
the kernels' own tolerance.  If the stand-in misreads a layout (GEGLU interleave, V^T padding, LayerNorm-fold statistics, GroupNorm
concat / modulation / lerp order, ...) the CPU tier's conclusions about the host logic would not transfer -- this file is what
makes them transfer."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from supir_amd import ops  # noqa: E402
from supir_amd import weights as Wt  # noqa: E402
from tests import torch_ops as TO  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def R(*shape, scale=1.0, dtype=BF, seed=[0]):
    seed[0] += 1
    g = torch.Generator(device="cpu").manual_seed(seed[0])
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(a, b, tol=4e-3):
    if isinstance(a, tuple):
        a, b = a[0], b[0]
    assert a.shape == b.shape and a.dtype == b.dtype
    e = rel_l2(a, b)
    assert e <= tol, e


@pytest.mark.parametrize("act", [0, 1, 3, 4])
def test_gemm_epilogues(act):
    M, N, K = 512, 640, 320
    a, w = R(2, M // 2, K), R(N, K, scale=K ** -0.5)
    bias, rb, res = R(N, dtype=torch.float32), R(2, N), R(2, M // 2, N)
    kw = dict(rowbias=rb, rows_per_batch=M // 2, residual=res, act=act, alpha=0.7)
    close(ops.gemm(a, w, bias, **kw), TO.gemm(a, w, bias, **kw))
    close(ops.gemm(a, w, bias, out_dtype=torch.float32), TO.gemm(a, w, bias, out_dtype=torch.float32), 1e-5 + 4e-3)


def test_gemm_geglu_interleave():
    M, K, n = 256, 320, 1280
    a = R(M, K)
    w, b = R(2 * n, K, scale=K ** -0.5), R(2 * n, dtype=torch.float32)
    wi, bi = Wt.interleave_geglu(w, b, 32)
    wi16, bi16 = Wt.interleave_geglu(w, b, 16)
    got = ops.gemm(a, wi, bi, act=2, alt16=(wi16, bi16))
    close(got, TO.gemm(a, wi, bi, act=2))
    want = (a.float() @ w.float().T + b)
    want = want[:, :n] * torch.nn.functional.gelu(want[:, n:])
    assert rel_l2(got, want) <= 4e-3


def test_layernorm_fold_producer_consumer_chain():
    M, C, N = 512, 640, 1280
    x0, res = R(2, M // 2, C), R(2, M // 2, C)
    w0, b0 = R(C, C, scale=C ** -0.5), R(C, dtype=torch.float32)
    gamma, beta = R(C, dtype=torch.float32) * 0.2 + 1.0, R(C, dtype=torch.float32) * 0.1
    w1, b1 = R(N, C, scale=C ** -0.5).float(), R(N, dtype=torch.float32)
    wp, colsum, bp = Wt.fold_layernorm(w1, b1, gamma, beta)
    outs = []
    for be in (ops, TO):
        x, st = be.gemm_ln(x0, w0, b0, residual=res, emit_stats=True)
        y = be.gemm_ln(x, wp, bp, ln=st, colsum=colsum, ln_eps=1e-5)
        yt = be.gemm_ln(x, wp, bp, ln=st, colsum=colsum, ln_eps=1e-5, trans=(2, M // 2, M // 2))
        fin = be.rowstats_finalize(st, C, 1e-5)
        y2 = be.gemm_ln(x, wp, bp, ln=fin, colsum=colsum, ln_eps=1e-5)
        outs.append((x, y, yt, y2))
    for a, b in zip(*outs):
        close(a, b)
    x = outs[0][0]
    want = torch.nn.functional.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w1.T + b1
    assert rel_l2(outs[0][1], want) <= 6e-3 and rel_l2(outs[1][1], want) <= 6e-3


def test_fused_qkv_and_transposed_projection():
    B, T, C = 2, 1024, 1280
    x = R(B, T, C)
    w, bias = R(3 * C, C, scale=C ** -0.5), R(3 * C, dtype=torch.float32)
    assert ops.gemm_qkv_supported(B * T, 3 * C, 2 * C, C, T)
    (qk, vt), (qk_r, vt_r) = ops.gemm_qkv(x, w, bias, B, T, 2 * C), TO.gemm_qkv(x, w, bias, B, T, 2 * C)
    close(qk, qk_r)
    close(vt, vt_r)
    xs = R(B, 77, C)
    close(ops.gemm_t(xs, w[:C].contiguous(), None, B, 77, 128), TO.gemm_t(xs, w[:C].contiguous(), None, B, 77, 128))


@pytest.mark.parametrize("kw", [dict(), dict(stride=2, pad=(1, 1)), dict(stride=2, pad=(0, 0), out_hw=(8, 8)), dict(upsample=True),
                                dict(act=1, alpha=0.5)])
def test_conv3x3_forms(kw):
    B, H, W, Ci, Co = 2, 16, 16, 128, 192
    x, w, b = R(B, H, W, Ci), R(Co, 3, 3, Ci, scale=(9 * Ci) ** -0.5), R(Co, dtype=torch.float32)
    close(ops.conv3x3(x, w, b, **kw), TO.conv3x3(x, w, b, **kw))
    if not kw:
        rb, res = R(B, Co), R(B, H, W, Co)
        close(ops.conv3x3(x, w, b, rowbias=rb, residual=res), TO.conv3x3(x, w, b, rowbias=rb, residual=res))
        wide = R(B, H, W, Ci + 64)      # a channel slice of a wider buffer as the input
        close(ops.conv3x3(wide[..., :Ci], w, b), TO.conv3x3(wide[..., :Ci], w, b))


@pytest.mark.parametrize("Tq,Tk,causal", [(1024, 1024, False), (256, 77, False), (77, 77, True)])
def test_flash_attention(Tq, Tk, causal):
    B, H = 2, 5
    C = H * 64
    qk = R(B, Tq, 2 * C) if Tq == Tk else None
    q = qk[:, :, :C] if qk is not None else R(B, Tq, C)
    k = qk[:, :, C:] if qk is not None else R(B, Tk, C)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, dtype=BF, device=DEV)
    vt[:, :, :Tk] = R(B, C, Tk)
    close(ops.flash_attn(q, k, vt, B, H, Tq, Tk, causal=causal), TO.flash_attn(q, k, vt, B, H, Tq, Tk, causal=causal), 6e-3)


def test_vae_attention_pieces():
    T_, C = 200, 512
    q, k = R(1, T_, C), R(1, T_, C)
    vt = torch.zeros(1, C, 224, dtype=BF, device=DEV)
    vt[:, :, :T_] = R(1, C, T_)
    close(ops.flash_attn_d512(q, k, vt, T_), TO.flash_attn_d512(q, k, vt, T_), 6e-3)
    s = R(64, 256, dtype=torch.float32, scale=3.0)
    close(ops.softmax_rows(s, 0.3, valid=200, dtype=BF), TO.softmax_rows(s, 0.3, valid=200, dtype=BF))


def test_groupnorm_forms():
    B, H, W, C1, C2 = 2, 8, 8, 640, 320
    x, x2 = R(B, H, W, C1), R(B, H, W, C2)
    C = C1 + C2
    g, b = R(C, dtype=torch.float32) * 0.2 + 1.0, R(C, dtype=torch.float32) * 0.1
    close(ops.groupnorm(x, g[:C1], b[:C1], 1e-5, silu=True), TO.groupnorm(x, g[:C1], b[:C1], 1e-5, silu=True))
    mg, mb, raw2 = R(B, H, W, C, scale=0.3), R(B, H, W, C, scale=0.3), R(B, H, W, C2)
    kw = dict(x2=x2, mod_g=mg, mod_b=mb, control_scale=0.6, x2raw=raw2)
    close(ops.groupnorm(x, g, b, 1e-5, **kw), TO.groupnorm(x, g, b, 1e-5, **kw))
    st, st_r = ops.groupnorm_stats(x), TO.groupnorm_stats(x)
    assert rel_l2(st, st_r) <= 1e-4
    n = H * W * (C1 // 32)
    given = torch.stack([st_r[..., 0] / n, st_r[..., 1] / n - (st_r[..., 0] / n) ** 2], dim=-1).contiguous()
    close(ops.groupnorm(x, g[:C1], b[:C1], 1e-6, given=given), TO.groupnorm(x, g[:C1], b[:C1], 1e-6, given=given))
    close(ops.layernorm(x, g[:C1], b[:C1], 1e-5), TO.layernorm(x, g[:C1], b[:C1], 1e-5))


def test_boundary_convs_and_elementwise():
    x = R(2, 4, 32, 32, dtype=torch.float32)
    w, b = R(320, 4, 3, 3, dtype=torch.float32, scale=0.2), R(320, dtype=torch.float32)
    add = R(2, 32, 32, 320)
    close(ops.conv3x3_smallcin(x, w, b, add=add), TO.conv3x3_smallcin(x, w, b, add=add))
    h, w9, b4 = R(2, 32, 32, 320), R(9, 4, 320, scale=0.02), R(4, dtype=torch.float32)
    assert rel_l2(ops.conv3x3_smallcout(h, w9, b4), TO.conv3x3_smallcout(h, w9, b4)) <= 4e-3
    pw, pb = R(8, 8, 1, 1, dtype=torch.float32), R(8, dtype=torch.float32)
    x8 = R(2, 8, 16, 16, dtype=torch.float32)
    assert rel_l2(ops.pointwise_nchw(x8, pw, pb, in_scale=1.7), TO.pointwise_nchw(x8, pw, pb, in_scale=1.7)) <= 1e-6
    lat, eps, ctr = (R(1, 4, 32, 32, dtype=torch.float32) for _ in range(3))
    for e in (eps, None):
        (xh, ni), (xh_r, ni_r) = ops.edm_step_pre(lat, e, 1.01, 0.3, 0.9, 2), TO.edm_step_pre(lat, e, 1.01, 0.3, 0.9, 2)
        assert rel_l2(xh, xh_r) <= 1e-6 and rel_l2(ni, ni_r) <= 1e-6
    net = R(2, 4, 32, 32, dtype=torch.float32)
    a = ops.edm_step_post(net, xh_r, ctr, -0.8, 1.0, 3.0, 0.2, 1.5, -0.4, 2)
    assert rel_l2(a, TO.edm_step_post(net, xh_r, ctr, -0.8, 1.0, 3.0, 0.2, 1.5, -0.4, 2)) <= 1e-5
    img = R(1, 3, 96, 80, dtype=torch.float32)
    (hi, lo), (hi_r, lo_r) = ops.wavelet_decomposition(img), TO.wavelet_decomposition(img)
    assert rel_l2(hi, hi_r) <= 1e-5 and rel_l2(lo, lo_r) <= 1e-5
