"""The product's HOST path on a box without a GPU: the module layer of supir_amd (derived weight layouts, LayerNorm folds, GEGLU
interleave, fused q|k|v stacks, channel-slice views, in-place residual streams, caches, ControlWrapper / samplers / VAE / tiled
VAE plumbing) with every kernel launch served by tests/torch_ops.py (plain torch, fp32), held to the golden vectors generated from
the REAL reference (tests/golden/golden_mini.pt, oracle/gen_golden.py).  What the `-m gpu` tier checks for the HIP kernels at bf16
tolerances (tests/test_model_gpu.py), this tier checks for everything above the C ABI at fp32 tolerances -- a wrong permutation,
fold, offset or cache key shows up here as an error of order one, with no GPU in sight.

tests/torch_ops.py itself is pinned to the kernels op by op in tests/test_torch_ops_vs_hip_gpu.py."""
import pytest
import torch

from tests import torch_ops
from tests.helpers import build_unet, build_vae, golden, golden_control, rel_l2, synth_tensor

B = 2
TOL = 2e-5     # fp32 summation-order noise of a different association of the same arithmetic (measured: 5e-8 .. 2e-6)


def T(name, shape, **kw):
    return synth_tensor(name, shape, **kw)


@pytest.fixture(scope="module")
def backend():
    with torch_ops.installed(fp32=True) as be:
        yield be


@pytest.fixture(scope="module")
def wrap(backend):
    w = build_unet(depth=(1, 1, 2), device="cpu")
    w.dtype = torch.float32
    return w


def _wrapper_inputs():
    x, lq = T("xt", (B, 4, 16, 16)), T("lq", (B, 4, 16, 16))
    y, ctx = T("vector", (B, 2816)), T("context", (B, 77, 2048))
    return x, torch.tensor([500, 37], dtype=torch.int64), {"crossattn": ctx, "vector": y, "control": lq}


def test_modules_vs_reference_golden(wrap):
    D, m = wrap.diffusion_model, golden()["modules"]
    emb, ctx = T("emb", (B, 1280)), T("context", (B, 77, 2048))
    x320, x640, x1280 = T("x320", (B, 320, 8, 8)), T("x640", (B, 640, 8, 8)), T("x1280", (B, 1280, 4, 4))
    hori1280, c1280 = T("hori1280", (B, 1280, 4, 4)), T("c1280", (B, 1280, 4, 4))
    c640, c320 = T("c640", (B, 640, 4, 4)), T("c320", (B, 320, 8, 8))
    P = D.project_modules
    cases = {
        "res.input_blocks.1.0": lambda: D.input_blocks[1][0](x320, emb),
        "res.input_blocks.4.0": lambda: D.input_blocks[4][0](x320, emb),
        "res.output_blocks.0.0": lambda: D.output_blocks[0][0](T("x2560", (B, 2560, 4, 4)), emb),
        "down.input_blocks.3.0": lambda: D.input_blocks[3][0](x320),
        "up.output_blocks.2.2": lambda: D.output_blocks[2][2](x1280),
        "st.input_blocks.4.1": lambda: D.input_blocks[4][1](x640, ctx),
        "st.middle_block.1": lambda: D.middle_block[1](x1280, ctx),
        "btb.input_blocks.7.1.0": lambda: D.input_blocks[7][1].transformer_blocks[0](T("tok1280", (B, 16, 1280)), ctx),
        "sft.11": lambda: P[11](c1280, x1280),
        "sft.10": lambda: P[10](c1280, x1280, hori1280),
        "sft.10.cs0.7": lambda: P[10](c1280, x1280, hori1280, control_scale=0.7),
        "sft.0": lambda: P[0](c320, x320, T("hori320", (B, 320, 8, 8))),
        "xattn.7": lambda: P[7](c640, x1280),
        "sft.11.cs0.6": lambda: P[11](c1280, x1280, control_scale=0.6),
        "xattn.7.cs0.6": lambda: P[7](c640, x1280, control_scale=0.6),
        "xattn.3": lambda: P[3](c320, x640),
    }
    assert set(cases) == set(m)
    with torch.no_grad():
        errs = {name: rel_l2(fn(), m[name]) for name, fn in cases.items()}
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, bad


@pytest.mark.parametrize("fused_qkv", [True, False])
def test_network_call_vs_reference_golden(wrap, backend, fused_qkv):
    """ControlWrapper.forward (GLVControl -> LightGLVUNet) vs the reference's own ControlWrapper output, control_scale 1 and 0.5,
    through both launch sequences of the self-attention projections (fused q|k|v and separate q|k + v^T)."""
    g = golden()
    x, t, cond = _wrapper_inputs()
    old = backend.PREFER_FUSED_QKV
    backend.PREFER_FUSED_QKV = fused_qkv
    try:
        with torch.no_grad():
            hs = wrap.control_model(x=cond["control"], timesteps=t, xt=x, context=cond["crossattn"], y=cond["vector"])
            eps = wrap(x, t, cond, 1.0)
            eps5 = wrap(x, t, cond, 0.5)
    finally:
        backend.PREFER_FUSED_QKV = old
    assert eps.dtype == torch.float32 and tuple(eps.shape) == (B, 4, 16, 16)
    assert rel_l2(eps, g["wrapper_eps"]) <= TOL and rel_l2(eps5, g["wrapper_eps_cs0.5"]) <= TOL
    assert len(hs) == 10
    full = golden_control()
    for i, h in enumerate(hs):
        d = g["control_digest"][i]
        assert list(h.shape) == d["shape"]
        f = h.float().contiguous().flatten()
        assert torch.allclose(torch.cat([f[:32], f[-32:]]), torch.cat([d["head"], d["tail"]]), rtol=1e-4, atol=1e-5 * d["std"])
        assert h.shape == full[i].shape and rel_l2(h, full[i]) <= TOL, i   # the reference's full tensors, all ten maps


def test_unfolded_transformer_path_matches_the_folded_one(wrap):
    """BasicTransformerBlock.forward (LayerNorm launches) and .forward_fused (norms folded into the consumer GEMMs) are the same
    function."""
    from supir_amd.modules import attention as A
    x, ctx = T("x1280", (B, 1280, 4, 4)), T("context", (B, 77, 2048))
    st = wrap.diffusion_model.middle_block[1]
    with torch.no_grad():
        a = st(x, ctx)
        A.FOLD_LAYERNORM = False
        try:
            b = st(x, ctx)
        finally:
            A.FOLD_LAYERNORM = True
    assert rel_l2(a, b) <= TOL


def test_embedding_schedule_serves_announced_calls_only(wrap):
    """ControlWrapper.prepare_schedule / select_step (round 4): a sampler that knows all timesteps of an image lets both networks
    build their time / label embedding projections for every step at once; an announced call reads its row out of that table and
    must equal the plain call, an un-announced call takes the plain path, a mismatching announcement is an error."""
    x, _, cond = _wrapper_inputs()
    t = torch.tensor([500, 500], dtype=torch.int64)
    with torch.no_grad():
        plain = wrap(x, t, cond, 1.0).clone()
        wrap.prepare_schedule([7, 500, 901], cond["vector"], control=cond["control"])
        try:
            assert wrap.diffusion_model._schedule["active"] is False
            wrap.select_step(1, expect_t=500)
            announced = wrap(x, t, cond, 1.0).clone()
            assert wrap.diffusion_model._schedule["active"] is True and not wrap._sched_armed      # consumed by that one call
            again = wrap(x, t, cond, 1.0).clone()                                                     # not announced: plain path
            assert wrap.diffusion_model._schedule["active"] is False
            wrap.select_step(2)
            other = wrap(x, torch.tensor([901, 901]), cond, 1.0).clone()
            with pytest.raises(ValueError):
                wrap.select_step(0, expect_t=500)
            with pytest.raises(IndexError):
                wrap.select_step(3)
        finally:
            wrap.end_schedule()
        ref_other = wrap(x, torch.tensor([901, 901]), cond, 1.0)
    assert rel_l2(announced, plain) <= 1e-6 and torch.equal(again, plain) and rel_l2(other, ref_other) <= 1e-6
    assert rel_l2(other, plain) > 1e-3          # another timestep really is another result
    assert wrap._sched is None


def test_sampler_2step_vs_reference_golden(wrap):
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, RestoreEDMSampler
    _, _, cond = _wrapper_inputs()
    ctx, y, lq = cond["crossattn"], cond["vector"], cond["control"]
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq[:1]}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq[:1]}
    den = DiscreteDenoiserWithControl()
    smp = RestoreEDMSampler(num_steps=2, s_churn=5, s_noise=1.01, restore_cfg=4.0, guider_config=LinearCFG(1.0, 4.0), device="cpu")
    smp.injected_step_noises = [T(f"eps{i}", (1, 4, 16, 16)) for i in range(2)]
    with torch.no_grad():
        out = smp(lambda i, s, cc, cs: den(wrap, i, s, cc, cs), T("noised_z", (1, 4, 16, 16)).clone(), cond=c, uc=uc,
                  x_center=T("x_center", (1, 4, 16, 16)), control_scale=1.0)
    assert rel_l2(out, golden()["sampler_2step"]) <= 5e-5


@pytest.fixture(scope="module")
def vae(backend):
    return build_vae("cpu")


def test_vae_vs_reference_golden(vae):
    from supir_amd.modules.vae import DiagonalGaussianDistribution
    g = golden()
    img = T("img", (1, 3, 64, 64), scale=0.5)
    with torch.no_grad():
        mom = vae.quant_conv(vae.denoise_encoder(img))
        z = DiagonalGaussianDistribution(mom).mode() * 0.13025
        xs1 = vae.decoder(vae.post_quant_conv(g["vae_z"], in_scale=1.0 / 0.13025))
        mom2 = vae.quant_conv(vae.encoder(g["vae_x_stage1"]))
    assert rel_l2(mom, g["vae_denoise_moments"]) <= TOL and rel_l2(z, g["vae_z"]) <= TOL
    assert rel_l2(xs1, g["vae_x_stage1"]) <= TOL and rel_l2(mom2, g["vae_moments2"]) <= TOL


def test_tiled_vae_vs_reference_golden(vae):
    """supir_amd/utils/tilevae.py (split, shape-group stacks, pooled statistics, crop + paste) vs the reference's VAEHook output."""
    from supir_amd.utils.tilevae import VAEHook
    g = golden()
    for net in (vae.denoise_encoder, vae.decoder):
        net.original_forward = net.forward
    with torch.no_grad():
        enc = VAEHook(vae.denoise_encoder, 64, is_decoder=False)(T("img_tiled", (1, 3, 192, 160), scale=0.5))
        dec = VAEHook(vae.decoder, 8, is_decoder=True)(T("z_tiled", (1, 4, 40, 32)))
    assert rel_l2(enc, g["tiled_enc_192x160_t64"]) <= 1e-4 and rel_l2(dec, g["tiled_dec_40x32_t8"]) <= 1e-4


def test_colour_fix_vs_reference_golden(backend):
    from supir_amd.utils.colorfix import wavelet_reconstruction
    out = wavelet_reconstruction(T("wa", (1, 3, 64, 64)), T("wb", (1, 3, 64, 64)))
    assert rel_l2(out, golden()["wavelet"]) <= TOL
