"""The fp32 service (libsupir_hip_f32.so, include/supir_hip_f32.h, supir_amd/ops_f32.py) -- what `--diff_dtype fp32` / `--ae_dtype fp32`
requests run on (reference: test.py:66-67; torch.autocast disables itself for float32, sgm/modules/diffusionmodules/wrappers.py:87;
SUPIR/models/SUPIR_model.py:13,41-69: fp32 is the constructor default, and the reference then computes in plain fp32).

Per operator: against a PyTorch float64 evaluation of the same op on the same fp32 operands (the kernels accumulate in fp32 in k order; an
fp32 library reference would carry its own rounding).  Bars: rel-L2 <= 2e-6, max-abs <= 2e-5 * max|ref| (measured values are ~1e-7).
Network level: one CFG-doubled ControlWrapper call and the VAE against the fp32 oracle <= 1e-5 (VERDICT r04 item 8's bar), with the bf16
build's error on the same inputs beside it (three orders of magnitude larger: proof of which arithmetic ran).
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from supir_amd import ops  # noqa: E402
from supir_amd import weights as Wt  # noqa: E402
from tests.helpers import build_unet, build_vae, rel_l2, synth_tensor  # noqa: E402

DEV = "cuda"
F32, F64 = torch.float32, torch.float64


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def check(out, ref, rel=2e-6, name=""):
    assert out.dtype == F32, out.dtype
    ref = ref.to(F64)
    o = out.to(F64)
    assert o.shape == ref.shape, (o.shape, ref.shape)
    assert torch.isfinite(o).all(), f"{name}: non-finite output"
    err = ((o - ref).norm() / (ref.norm() + 1e-30)).item()
    mx = (o - ref).abs().max().item()
    assert err <= rel, f"{name}: rel-L2 {err:.3e} > {rel}"
    assert mx <= 10 * rel * ref.abs().max().item() + 1e-7, f"{name}: max-abs {mx:.3e}"
    return err


def _act(y, act):
    if act == 1:
        return F.silu(y)
    if act == 3:
        return F.gelu(y)
    if act == 4:
        return y * torch.sigmoid(1.702 * y)
    return y


def test_f32_library_is_what_fp32_operands_reach():
    from supir_amd import _lib
    lib32 = _lib.load(F32)
    assert lib32.supir_elem_type() == b"f32" and lib32 is not _lib.load()
    a, w = rnd(64, 64), rnd(64, 64, seed=1)
    out = ops.gemm(a, w)
    assert out.dtype == F32
    check(out, a.double() @ w.double().T, name="gemm 64^3")
    with pytest.raises(AssertionError):
        ops.gemm(a, w.to(torch.bfloat16))   # mixed element types are refused, never converted behind the caller's back
    assert not ops.has_fused(F32) and ops.has_fused(torch.bfloat16)


# both tiles (64 x 64 below M = 1024 or N = 96; 128 x 128 above), ragged edges in every dimension, K not a multiple of 4 (scalar loads)
@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (2048, 1280, 5120), (100, 70, 36), (1, 1280, 320), (1030, 130, 67), (77, 2048, 2048),
                                   (4096, 64, 4096), (2, 2816, 1280)])
def test_gemm_plain_f32(M, N, K):
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5, seed=1)
    check(ops.gemm(a, w), a.double() @ w.double().T, name=f"gemm {M}x{N}x{K}")


def test_gemm_epilogues_f32():
    M, N, K, B = 2048, 640, 320, 2
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5, seed=1)
    bias, rb, res = rnd(N, seed=2), rnd(B, N, seed=3), rnd(M, N, seed=4)
    base = a.double() @ w.double().T + bias.double()
    for act in (0, 1, 3, 4):
        check(ops.gemm(a, w, bias, act=act), _act(base, act), name=f"act {act}")
    full = _act(base + rb.double().repeat_interleave(M // B, 0), 1) * 0.7 + res.double()
    check(ops.gemm(a, w, bias, rowbias=rb, rows_per_batch=M // B, residual=res, act=1, alpha=0.7), full, name="all epilogue terms")
    # in place on the residual (x += f(x)), strided input view, strided output view
    big = rnd(M, 2 * K, seed=5)
    x = res.clone()
    ops.gemm(big[:, K:], w, bias, residual=x, out=x)
    check(x, big[:, K:].double() @ w.double().T + bias.double() + res.double(), name="in place / strided A")
    wide = torch.zeros(M, 2 * N, device=DEV)
    ops.gemm(a, w, bias, out=wide[:, N:])
    check(wide[:, N:].contiguous(), base, name="strided C")
    assert wide[:, :N].abs().max().item() == 0
    out, part = ops.gemm(a, w, bias, rows_per_batch=M // B, gn_part=True)     # no GroupNorm partials in fp32: the consumer runs its own pass
    assert part is None
    check(out, base, name="gn_part request")


def test_gemm_t_and_geglu_f32():
    B, T, K, N = 2, 77, 2048, 640
    Tp = 128
    a, w, bias = rnd(B * T, K), rnd(N, K, scale=K ** -0.5, seed=1), rnd(N, seed=2)
    vt = ops.gemm_t(a, w, bias, B, T, Tp)
    ref = (a.double() @ w.double().T + bias.double()).view(B, T, N).transpose(1, 2)
    check(vt[:, :, :T].contiguous(), ref, name="gemm_t")
    assert vt.shape == (B, N, Tp) and vt[:, :, T:].abs().max().item() == 0
    M, K, N2 = 512, 320, 2560
    a = rnd(M, K)
    w, b = rnd(N2, K, scale=K ** -0.5, seed=3), rnd(N2, seed=4)
    wi, bi = Wt.interleave_geglu(w, b)            # the layout the modules hand to ops.gemm(act=2) in every element type
    proj = a.double() @ w.double().T + b.double()
    check(ops.gemm(a, wi, bi, act=2), proj[:, :N2 // 2] * F.gelu(proj[:, N2 // 2:]), name="geglu")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pad,up,out_hw", [
    (2, 32, 32, 320, 320, 1, (1, 1), False, None), (2, 32, 32, 640, 320, 1, (1, 1), False, None), (2, 32, 32, 320, 320, 2, (1, 1), False, None),
    (2, 16, 16, 640, 640, 1, (1, 1), True, None), (1, 64, 64, 128, 128, 2, (0, 0), False, (32, 32)), (1, 30, 34, 128, 256, 1, (1, 1), False, None),
    (2, 32, 32, 4, 320, 1, (1, 1), False, None), (1, 33, 31, 3, 128, 1, (1, 1), False, None), (1, 64, 64, 128, 3, 1, (1, 1), False, None),
    (1, 31, 33, 130, 8, 2, (0, 0), False, (15, 16))])
def test_conv3x3_f32(B, H, W, Cin, Cout, stride, pad, up, out_hw):
    x = rnd(B, H, W, Cin)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1)
    bias = rnd(Cout, seed=2)
    wk = w.permute(0, 2, 3, 1).contiguous()
    xr = x.double().permute(0, 3, 1, 2)
    if up:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if out_hw is not None:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.double(), bias.double(), stride=stride, padding=0)
    else:
        ref = F.conv2d(xr, w.double(), bias.double(), stride=stride, padding=1)
    assert out_hw is None or tuple(ref.shape[2:]) == tuple(out_hw)
    ref = ref.permute(0, 2, 3, 1)
    check(ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw), ref, name="conv")
    OH, OW = ref.shape[1:3]
    rb, res = rnd(B, Cout, seed=3), rnd(B, OH, OW, Cout, seed=4)
    full = F.silu(ref + rb.double()[:, None, None, :]) * 0.5 + res.double()
    check(ops.conv3x3(x, wk, bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw, rowbias=rb, residual=res, act=1, alpha=0.5), full,
          name="conv + epilogue")


def test_boundary_convs_f32():
    x = rnd(2, 4, 32, 32)
    w, b = rnd(320, 4, 3, 3, scale=1 / 6, seed=1), rnd(320, seed=2)
    add = rnd(2, 32, 32, 320, seed=3)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    check(ops.conv3x3_smallcin(x, w, b, dtype=F32), ref, name="smallcin")
    check(ops.conv3x3_smallcin(x, w, b, add=add), ref + add.double(), name="smallcin + add")
    h = rnd(1, 40, 40, 128)
    w3, b3 = rnd(3, 128, 3, 3, scale=1 / 34, seed=4), rnd(3, seed=5)
    with Wt.compute_dtype(F32):               # the derived layouts follow the scope's element type
        w9 = Wt.conv3x3_w9(w3)
    out = ops.conv3x3_smallcout(h, w9, b3)
    assert out.shape == (1, 3, 40, 40) and out.is_contiguous()
    check(out, F.conv2d(h.double().permute(0, 3, 1, 2), w3.double(), b3.double(), padding=1), name="smallcout")


def _gn_ref(x, gamma, beta, eps):
    B, C = x.shape[0], x.shape[-1]
    y = F.group_norm(x.double().reshape(B, -1, C).transpose(1, 2), 32, gamma.double(), beta.double(), eps)
    return y.transpose(1, 2).reshape(x.shape)


def test_groupnorm_layernorm_softmax_f32():
    B, H, W = 2, 16, 16
    x = rnd(B, H, W, 320) * 3 + 1.5
    g, b = rnd(320, seed=1) * 0.2 + 1, rnd(320, seed=2) * 0.2
    check(ops.groupnorm(x, g, b, 1e-5), _gn_ref(x, g, b, 1e-5), name="gn")
    check(ops.groupnorm(x, g, b, 1e-6, silu=True), F.silu(_gn_ref(x, g, b, 1e-6)), name="gn + silu")
    y = x.clone()
    ops.groupnorm(y, g, b, 1e-5, silu=True, out=y)
    check(y, F.silu(_gn_ref(x, g, b, 1e-5)), name="gn in place")
    # ZeroSFT tail (SUPIR/modules/SUPIR_v0.py:91-113): concat [h_ori (640) | h (1280)] = 1920 channels, 60 per group: the seam at 640 falls
    # INSIDE group 10; modulation; control-scale lerp against the raw concat
    x1, x2, x2raw = rnd(B, H, W, 640, seed=3), rnd(B, H, W, 1280, seed=4), rnd(B, H, W, 1280, seed=5)
    C = 1920
    g, b = rnd(C, seed=6) * 0.2 + 1, rnd(C, seed=7) * 0.2
    gb = rnd(B, H, W, 2 * C, seed=8)
    cat = torch.cat([x1, x2], -1)
    mod = _gn_ref(cat, g, b, 1e-5) * (gb[..., :C].double() + 1) + gb[..., C:].double()
    check(ops.groupnorm(x1, g, b, 1e-5, x2=x2, mod_g=gb[..., :C], mod_b=gb[..., C:]), mod, name="gn concat + modulation")
    raw = torch.cat([x1, x2raw], -1).double()
    check(ops.groupnorm(x1, g, b, 1e-5, x2=x2, mod_g=gb[..., :C], mod_b=gb[..., C:], control_scale=0.7, x2raw=x2raw), mod * 0.7 + raw * 0.3,
          name="gn concat + modulation + lerp")
    # a large group (VAE: 128 channels at 256^2 pixels = 262144 elements per group, several row slices)
    xv = rnd(1, 256, 256, 128, seed=9) * 2 - 0.5
    gv, bv = rnd(128, seed=10) * 0.1 + 1, rnd(128, seed=11) * 0.1
    check(ops.groupnorm(xv, gv, bv, 1e-6, silu=True), F.silu(_gn_ref(xv, gv, bv, 1e-6)), name="gn large")
    t = rnd(300, 1280) * 2 + 0.3
    g, b = rnd(1280, seed=12) * 0.2 + 1, rnd(1280, seed=13) * 0.2
    check(ops.layernorm(t, g, b), F.layer_norm(t.double(), (1280,), g.double(), b.double(), 1e-5), name="layernorm")
    s = rnd(200, 128) * 4
    p = ops.softmax_rows(s, 0.3, valid=77, dtype=F32)
    check(p[:, :77].contiguous(), torch.softmax(s[:, :77].double() * 0.3, -1), name="softmax")
    assert p[:, 77:].abs().max().item() == 0


@pytest.mark.parametrize("B,H,Tq,Tk", [(2, 5, 1024, 1024), (2, 10, 256, 77), (1, 20, 100, 300)])
def test_attention_f32(B, H, Tq, Tk):
    C = H * 64
    Tp = (Tk + 63) // 64 * 64
    qk = rnd(B, Tq, 2 * C)                      # q read as a row-strided view of a wider buffer, as the self-attention path does
    q = qk[:, :, :C]
    k, v = rnd(B, Tk, C, seed=1), rnd(B, Tk, C, seed=2)
    vt = torch.zeros(B, C, Tp, device=DEV)
    vt[:, :, :Tk] = v.transpose(1, 2)
    out = ops.flash_attn(q, k, vt, B, H, Tq, Tk)
    qh, kh, vh = (t.double().reshape(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, Tq, C)
    check(out, ref, name="attention d64")


def test_attention_causal_f32():
    """The text towers' causal self-attention (77 tokens padded to 128 keys): masked probabilities are exact zeros."""
    B, H, T = 2, 12, 77
    C, Tp = H * 64, 128
    qk = rnd(B, T, 2 * C)
    v = rnd(B, T, C, seed=2)
    vt = torch.zeros(B, C, Tp, device=DEV)
    vt[:, :, :T] = v.transpose(1, 2)
    out = ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, B, H, T, T, causal=True)
    qh, kh, vh = (t.double().reshape(B, T, H, 64).transpose(1, 2) for t in (qk[:, :, :C], qk[:, :, C:], v))
    sc = qh @ kh.transpose(-1, -2) * 0.125
    sc = sc.masked_fill(torch.ones(T, T, dtype=torch.bool, device=DEV).triu(1), float("-inf"))
    check(out, (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, T, C), name="causal attention")


def test_attention_d512_f32():
    B, T = 2, 324
    Tp = (T + 63) // 64 * 64
    q, k, v = rnd(B, T, 512), rnd(B, T, 512, seed=1), rnd(B, T, 512, seed=2)
    vt = torch.zeros(B, 512, Tp, device=DEV)
    vt[:, :, :T] = v.transpose(1, 2)
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) * 512 ** -0.5, -1) @ v.double()
    check(ops.flash_attn_d512(q, k, vt, T), ref, name="attention d512")


# ------------------------------------------------------------------------------------------------- network level
def _record(name, **vals):
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    print(f"[parity-fp32] {name}: " + ", ".join(f"{k}={v:.4g}" for k, v in vals.items()))
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_fp32.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = vals
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope="module")
def mini():
    return build_unet(depth=(1, 1, 2), device=DEV)


def _sd_of(wrap):
    sd = {}
    for pfx, mod in (("model.diffusion_model.", wrap.diffusion_model), ("model.control_model.", wrap.control_model)):
        for k, v in mod.state_dict().items():
            sd[pfx + k] = v
    return sd


@pytest.mark.parametrize("control_scale", [1.0, 0.8])
def test_network_call_fp32_vs_fp32_oracle(mini, control_scale):
    """One CFG-doubled ControlWrapper call (reduced depth, real widths, latent 32^2) with dtype = torch.float32 against the fp32 oracle:
    <= 1e-5; no warning (the request is honoured); graph replay requested -> still the eager fp32 launches; and the bf16 path is untouched
    by the excursion (its layouts are rebuilt, its result is bitwise what it was)."""
    import warnings
    from oracle import supir_oracle as O
    B, L = 2, 32
    x = synth_tensor("xt32", (B, 4, L, L)).to(DEV)
    cond = {"crossattn": synth_tensor("context", (B, 77, 2048)).to(DEV), "vector": synth_tensor("vector", (B, 2816)).to(DEV),
            "control": synth_tensor("lq32", (B, 4, L, L)).to(DEV)}
    t = torch.tensor([500, 37], dtype=torch.int64, device=DEV)
    with torch.no_grad():
        ref = O.control_wrapper(_sd_of(mini), x, t, cond, control_scale)
        bf_before = mini(x, t, cond, control_scale).clone()
        mini.dtype = F32
        try:
            assert mini.effective_dtype == F32
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                out32 = mini(x, t, cond, control_scale).clone()
            assert not [r for r in rec if issubclass(r.category, RuntimeWarning)]
            mini.enable_graph(True)
            g32 = mini(x, t, cond, control_scale).clone()
            assert not mini._graphs
        finally:
            mini.enable_graph(False)
            mini.dtype = torch.bfloat16
        bf_after = mini(x, t, cond, control_scale)
    e32, ebf = rel_l2(out32, ref), rel_l2(bf_before, ref)
    _record(f"network_call_mini_latent32_cs{control_scale:g}", fp32_vs_oracle=e32, bf16_vs_oracle=ebf, max_abs=(out32 - ref).abs().max().item())
    assert out32.dtype == F32 and torch.isfinite(out32).all()
    assert e32 <= 1e-5, e32
    assert ebf >= 100 * e32
    assert torch.equal(g32, out32) and torch.equal(bf_after, bf_before)


def test_prepared_schedule_in_fp32_after_a_bf16_image(mini):
    """The per-image embedding schedule (ControlWrapper.prepare_schedule / select_step) in an fp32 scope, AFTER a bf16 image was sampled with
    its own schedule at the same batch size and ANOTHER LQ latent: the fp32 steps must read the fp32 tables (and the fp32 hint of THEIR
    control input), i.e. equal the plain fp32 calls.  (Regression: the table was once looked up outside the compute-dtype scope, which found
    the bf16 image's table -- and its stale bf16 input_hint_block output -- for an fp32 step.)"""
    B, L = 2, 32
    x = synth_tensor("xt32", (B, 4, L, L)).to(DEV)
    cond = {"crossattn": synth_tensor("context", (B, 77, 2048)).to(DEV), "vector": synth_tensor("vector", (B, 2816)).to(DEV),
            "control": synth_tensor("lq32", (B, 4, L, L)).to(DEV)}
    other = dict(cond, control=synth_tensor("lq32.other", (B, 4, L, L)).to(DEV))
    tt = lambda tv: torch.full((B,), tv, dtype=torch.int64, device=DEV)
    try:
        with torch.no_grad():
            mini.prepare_schedule([900, 40], other["vector"], control=other["control"])      # a bf16 image, another LQ latent
            mini.select_step(0, expect_t=900)
            mini(x, tt(900), other, 1.0)
            mini.end_schedule()
            mini.dtype = F32
            plain = [mini(x, tt(tv), cond, 1.0).clone() for tv in (900, 40)]
            mini.prepare_schedule([900, 40], cond["vector"], control=cond["control"])
            for i, tv in enumerate((900, 40)):
                mini.select_step(i, expect_t=tv)
                out = mini(x, tt(tv), cond, 1.0)
                assert mini.diffusion_model._schedule["cdt"] == F32 and mini.control_model._schedule["hint"].dtype == F32
                assert rel_l2(out, plain[i]) <= 2e-6, (i, rel_l2(out, plain[i]))
            # and the fp16 build, same sequence (its table, its hint)
            mini.end_schedule()
            mini.dtype = torch.float16
            p16 = mini(x, tt(900), cond, 1.0).clone()
            mini.prepare_schedule([900, 40], cond["vector"], control=cond["control"])
            mini.select_step(0, expect_t=900)
            o16 = mini(x, tt(900), cond, 1.0)
            assert mini.diffusion_model._schedule["cdt"] == torch.float16 and mini.control_model._schedule["hint"].dtype == torch.float16
            assert rel_l2(o16, p16) <= 2e-3, rel_l2(o16, p16)
    finally:
        mini.end_schedule()
        mini.dtype = torch.bfloat16


def test_vae_fp32_vs_fp32_oracle():
    """encoder -> quant_conv and post_quant_conv -> decoder (sgm/models/autoencoder.py:282-321, model.py:482-743) in an fp32 scope against
    the oracle, 128^2 pixels (mid-block attention over 256 tokens: the materialised-score form) and, for the decoder, a 40 x 24 latent."""
    from oracle import supir_oracle as O
    vae = build_vae(device=DEV)
    sd = {"first_stage_model." + k: v for k, v in vae.state_dict().items()}
    img = synth_tensor("vae.img", (1, 3, 128, 128), scale=0.5).clamp(-1, 1).to(DEV)
    z = synth_tensor("vae.z", (1, 4, 40, 24)).to(DEV)
    with torch.no_grad():
        ref_m = O.vae_moments(sd, img)
        ref_d = O.vae_decode(sd, z)
        with Wt.compute_dtype(F32):
            assert Wt.cdt() == F32
            m32 = vae.quant_conv(vae.encoder(img))
            d32 = vae.decoder(vae.post_quant_conv(z))
        mbf = vae.quant_conv(vae.encoder(img))
    em, ed, ebf = rel_l2(m32, ref_m), rel_l2(d32, ref_d), rel_l2(mbf, ref_m)
    _record("vae_128px", moments_fp32_vs_oracle=em, decode_fp32_vs_oracle=ed, moments_bf16_vs_oracle=ebf)
    assert m32.dtype == F32 and d32.dtype == F32 and d32.shape == (1, 3, 320, 192)
    assert em <= 1e-5 and ed <= 1e-5, (em, ed)
    assert ebf >= 100 * em


def test_tiled_vae_fp32_vs_oracle():
    """`--use_tile_vae --ae_dtype fp32` (test.py:65-67): the VAEHook task queue with pooled GroupNorm statistics (SUPIR/utils/tilevae.py:524-553,
    610-640) in an fp32 scope -- per-tile (sum, sum of squares) in fp64 from supir_f32_groupnorm_stats, the pooled (mean, variance) handed
    back to supir_f32_groupnorm -- against the oracle's tiled forward: decoder on a 48 x 40 latent in 16-latent tiles, encoder on a
    256 x 320 image in 128-px tiles."""
    from oracle import supir_oracle as O
    from supir_amd.utils.tilevae import VAEHook
    vae = build_vae(device=DEV)
    sd = dict(vae.state_dict())
    z = synth_tensor("vae.tz", (1, 4, 48, 40)).to(DEV)
    img = synth_tensor("vae.timg", (1, 3, 256, 320), scale=0.5).clamp(-1, 1).to(DEV)
    x = rnd(3, 24, 24, 128) * 2 + 0.5
    with torch.no_grad():
        sums = ops.groupnorm_stats(x)
        assert sums.dtype == F64 and sums.shape == (3, 32, 2)
        xg = x.double().reshape(3, 576, 32, 4)
        assert torch.allclose(sums[..., 0], xg.sum(dim=(1, 3)), rtol=1e-12) and torch.allclose(sums[..., 1], (xg * xg).sum(dim=(1, 3)), rtol=1e-12)
        given = torch.stack([rnd(3, 32, seed=1) * 0.1, rnd(3, 32, seed=2).abs() + 0.5], -1).contiguous()
        g, b = rnd(128, seed=3) * 0.2 + 1, rnd(128, seed=4) * 0.2
        ref = (xg - given[:, None, :, None, 0].double()) / torch.sqrt(given[:, None, :, None, 1].double() + 1e-6)
        ref = ref.reshape(3, 24, 24, 128) * g.double() + b.double()
        check(ops.groupnorm(x, g, b, 1e-6, given=given), ref, name="gn with pooled statistics")
        with Wt.compute_dtype(F32):
            dec = VAEHook(vae.decoder, 16, is_decoder=True)(z)
            enc = VAEHook(vae.denoise_encoder, 128, is_decoder=False)(img)
        ref_dec = O.vae_tiled_forward(sd, z, "decoder.", 16, True)
        ref_enc = O.vae_tiled_forward(sd, img, "denoise_encoder.", 128, False)
    e_dec, e_enc = rel_l2(dec, ref_dec), rel_l2(enc, ref_enc)
    _record("tiled_vae_fp32", decoder=e_dec, encoder=e_enc)
    assert dec.dtype == F32 and tuple(dec.shape) == (1, 3, 384, 320) and tuple(enc.shape) == (1, 8, 32, 40)
    assert e_dec <= 1e-5 and e_enc <= 1e-5, (e_dec, e_enc)


def test_text_towers_fp32_vs_oracle():
    """The conditioner under `--ae_dtype fp32` (the reference runs it under autocast(ae_dtype), SUPIR_model.py:165: plain fp32 then): CLIP-L
    hidden_states[11] and OpenCLIP bigG penultimate / pooled at full size in an fp32 scope against the oracle."""
    from oracle import cond_oracle as CO
    from supir_amd.modules import conditioner as C
    from tests.test_conditioner import _fill, _tokens
    tok = _tokens(2, seed=3)
    with torch.device(DEV):
        cl = C.FrozenCLIPEmbedder(layer="hidden", layer_idx=11)
        g = C.FrozenOpenCLIPEmbedder2(arch="ViT-bigG-14", layer="penultimate", always_return_pooled=True, legacy=False)
    _fill(cl, "conditioner.embedders.0.", DEV)
    _fill(g, "conditioner.embedders.1.", DEV)
    with torch.no_grad():
        with Wt.compute_dtype(F32):
            z = cl(tok)
            pen, pooled = g(tok)
        zbf = cl(tok)
        ref_z = CO.clip_l_hidden(cl.state_dict(), tok.to(DEV), p="transformer.text_model.")
        ref_pen, ref_pool = CO.openclip_g_penultimate_pooled(g.state_dict(), tok.to(DEV), p="model.")
    e = dict(clip_l_hidden11=rel_l2(z, ref_z), bigg_penultimate=rel_l2(pen, ref_pen), bigg_pooled=rel_l2(pooled, ref_pool),
             clip_l_bf16=rel_l2(zbf, ref_z))
    _record("text_towers", **e)
    assert max(e["clip_l_hidden11"], e["bigg_penultimate"], e["bigg_pooled"]) <= 1e-5, e
    assert e["clip_l_bf16"] >= 100 * e["clip_l_hidden11"]
