"""HIP product path (module layer -> C ABI -> gfx950 kernels) against the golden vectors of the REAL reference and
against the oracle, on the GPU.

Tolerances (SURVEY.md 8(d) / BASELINE.md section 5): the reference's own bf16-autocast path differs from its fp32 path by
rel-L2 1.5e-2 on the raw network output; a from-scratch bf16 path cannot be closer to the fp32 golden than that.
Bars: single modules rel-L2 <= 1e-2, whole network call (ControlWrapper eps) <= 2e-2, VAE <= 2e-2.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import (build_unet, build_vae, golden, golden_control, manifest, rel_l2, synth_sd, synth_tensor)  # noqa: E402

DEV = "cuda"
B = 2


@pytest.fixture(scope="module")
def wrap():
    return build_unet(depth=(1, 1, 2), device=DEV)


@pytest.fixture(scope="module")
def g():
    return golden()


def T(name, shape, **kw):
    return synth_tensor(name, shape, **kw).to(DEV)


def test_state_dict_keys_match_reference_manifest(wrap):
    man = manifest("mini")
    ours = {"model.diffusion_model." + k: list(v.shape) for k, v in wrap.diffusion_model.state_dict().items()}
    ours.update({"model.control_model." + k: list(v.shape) for k, v in wrap.control_model.state_dict().items()})
    ref = {k: v for k, v in man.items() if k.startswith("model.")}
    assert ours == ref


def test_modules_vs_reference_golden(wrap, g):
    D = wrap.diffusion_model
    m = g["modules"]
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
    errs = {}
    with torch.no_grad():
        for name, fn in cases.items():
            out = fn()
            assert tuple(out.shape) == tuple(m[name].shape), name
            errs[name] = rel_l2(out, m[name])
    print({k: f"{v:.2e}" for k, v in errs.items()})
    # single modules: <= 1e-2; transformer stacks (depth >= 1, ~15 GEMMs + 2 softmaxes deep) <= 1.5e-2
    bad = {k: v for k, v in errs.items() if not v <= (1.5e-2 if k.startswith(("st.", "btb.")) else 1e-2)}
    assert not bad, bad


def _wrapper_inputs():
    x, lq = T("xt", (B, 4, 16, 16)), T("lq", (B, 4, 16, 16))
    y, ctx = T("vector", (B, 2816)), T("context", (B, 77, 2048))
    t = torch.tensor([500, 37], dtype=torch.int64, device=DEV)
    return x, t, {"crossattn": ctx, "vector": y, "control": lq}


def test_control_features_and_wrapper_vs_reference_golden(wrap, g):
    x, t, cond = _wrapper_inputs()
    with torch.no_grad():
        hs = wrap.control_model(x=cond["control"], timesteps=t, xt=x, context=cond["crossattn"], y=cond["vector"])
        assert len(hs) == 10
        for i, (h, d) in enumerate(zip(hs, g["control_digest"])):
            assert list(h.shape) == d["shape"]
            assert abs(h.float().std().item() - d["std"]) <= 2e-2 * d["std"]
            # element-wise: the first / last 32 values of the reference's feature map (flattened NCHW order), rel-L2 on the 64
            # stored values against the map's own scale (tests/test_parity_production_gpu.py checks all 10 maps in full against
            # the oracle at latent 64^2)
            f = h.float().contiguous().flatten().cpu()
            got, want = torch.cat([f[:32], f[-32:]]), torch.cat([d["head"], d["tail"]])
            assert (got - want).norm().item() <= 2.5e-2 * d["std"] * 8.0, (i, (got - want).norm().item(), d["std"])   # 8 = sqrt(64)
        # all ten maps against the reference's FULL tensors (round 4: golden_control.pt)
        errs = [rel_l2(h, full) for h, full in zip(hs, golden_control())]
        print("control features vs reference (full tensors):", [f"{e:.2e}" for e in errs])
        assert max(errs) <= 1.5e-2, errs
        eps = wrap(x, t, cond, 1.0)
        assert eps.dtype == torch.float32 and tuple(eps.shape) == (B, 4, 16, 16)
        e1 = rel_l2(eps, g["wrapper_eps"])
        e2 = rel_l2(wrap(x, t, cond, 0.5), g["wrapper_eps_cs0.5"])
    print(f"wrapper eps rel-L2 vs fp32 reference: {e1:.3e} (cs=1), {e2:.3e} (cs=0.5)")
    assert e1 <= 2.5e-2 and e2 <= 2.5e-2


def _oracle_sd(wrap):
    sd = {}
    for pfx, mod in (("model.diffusion_model.", wrap.diffusion_model), ("model.control_model.", wrap.control_model)):
        for k, v in mod.state_dict().items():
            sd[pfx + k] = v
    return sd


def _bf16_floor(sd, x, t, cond, cs=1.0):
    """How far the REFERENCE-STYLE bf16 path (ATen ops under torch.autocast, exactly what ControlWrapper does at
    wrappers.py:87) lands from the fp32 oracle on the same weights / inputs: the noise floor for any bf16 path."""
    from oracle import supir_oracle as O
    with torch.no_grad():
        ref = O.control_wrapper(sd, x, t, cond, cs)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lo = O.control_wrapper(sd, x, t, cond, cs)
    return ref, rel_l2(lo, ref)


def test_wrapper_is_bitwise_reproducible(wrap):
    x, t, cond = _wrapper_inputs()
    with torch.no_grad():
        a = wrap(x, t, cond, 1.0).clone()
        b = wrap(x, t, cond, 1.0).clone()
    assert torch.equal(a, b)


def test_wrapper_vs_oracle_with_measured_bf16_floor(wrap):
    x, t, cond = _wrapper_inputs()
    sd = _oracle_sd(wrap)
    for cs in (1.0, 0.5):
        ref, floor = _bf16_floor(sd, x, t, cond, cs)
        with torch.no_grad():
            e = rel_l2(wrap(x, t, cond, cs), ref)
        print(f"mini wrapper cs={cs}: HIP bf16 vs fp32 oracle {e:.3e}; ATen-autocast bf16 vs fp32 oracle (floor) {floor:.3e}")
        assert e <= max(2e-2, 1.5 * floor)


def test_wrapper_with_the_references_erf_gelu(wrap, monkeypatch):
    """SUPIR_EXACT_GELU=1 / ops.EXACT_GELU: every GEGLU launch of the network call takes SUPIR_ACT_GEGLU_ERF -- the reference's F.gelu
    (sgm/modules/attention.py:89-91) instead of the fitted form.  Same bar against the fp32 oracle; the two runs differ (the switch reaches
    the kernels) by less than the bf16 floor."""
    from supir_amd import ops
    x, t, cond = _wrapper_inputs()
    sd = _oracle_sd(wrap)
    ref, floor = _bf16_floor(sd, x, t, cond, 1.0)
    with torch.no_grad():
        fitted = wrap(x, t, cond, 1.0).clone()
        monkeypatch.setattr(ops, "EXACT_GELU", True)
        exact = wrap(x, t, cond, 1.0).clone()
        assert torch.equal(exact, wrap(x, t, cond, 1.0))
    monkeypatch.setattr(ops, "EXACT_GELU", False)
    e_fit, e_erf, d = rel_l2(fitted, ref), rel_l2(exact, ref), rel_l2(exact, fitted)
    print(f"mini wrapper: fitted GELU vs fp32 oracle {e_fit:.3e}; erf GELU {e_erf:.3e}; erf vs fitted {d:.3e}; ATen-bf16 floor {floor:.3e}")
    assert e_erf <= max(2e-2, 1.5 * floor) and 0 < d < floor


def test_wrapper_graph_replay_matches_eager(wrap):
    x, t, cond = _wrapper_inputs()
    with torch.no_grad():
        ref = wrap(x, t, cond, 1.0).clone()
        wrap.enable_graph(True)
        try:
            a = wrap(x, t, cond, 1.0).clone()
            x2 = x * 0.5
            b = wrap(x2, t, cond, 1.0).clone()
            wrap.enable_graph(False)
            b_ref = wrap(x2, t, cond, 1.0)
        finally:
            wrap.enable_graph(False)
    assert torch.equal(a, ref)
    assert torch.equal(b, b_ref)


def test_sampler_2step_vs_reference_golden(wrap, g):
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, RestoreEDMSampler
    _, _, cond = _wrapper_inputs()
    ctx, y, lq = cond["crossattn"], cond["vector"], cond["control"]
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq[:1]}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq[:1]}
    den = DiscreteDenoiserWithControl().to(DEV)
    smp = RestoreEDMSampler(num_steps=2, s_churn=5, s_noise=1.01, restore_cfg=4.0, guider_config=LinearCFG(1.0, 4.0), device=DEV)
    noises = iter([T(f"eps{i}", (1, 4, 16, 16)) for i in range(2)])
    orig = torch.randn_like
    torch.randn_like = lambda t_, **kw: next(noises).to(t_)
    try:
        with torch.no_grad():
            out = smp(lambda i, s, cc, cs: den(wrap, i, s, cc, cs), T("noised_z", (1, 4, 16, 16)).clone(), cond=c, uc=uc,
                      x_center=T("x_center", (1, 4, 16, 16)), control_scale=1.0)
    finally:
        torch.randn_like = orig
    e = rel_l2(out, g["sampler_2step"])
    print(f"2-step sampler rel-L2 vs fp32 reference: {e:.3e}")
    assert e <= 3e-2


def test_vae_vs_reference_golden(g):
    vae = build_vae(DEV)
    img = T("img", (1, 3, 64, 64), scale=0.5)
    from supir_amd.modules.vae import DiagonalGaussianDistribution
    with torch.no_grad():
        mom = vae.quant_conv(vae.denoise_encoder(img))
        e_m = rel_l2(mom, g["vae_denoise_moments"])
        z = DiagonalGaussianDistribution(mom).mode() * 0.13025
        xs1 = vae.decoder(vae.post_quant_conv(g["vae_z"].to(DEV), in_scale=1.0 / 0.13025))
        e_d = rel_l2(xs1, g["vae_x_stage1"])
        mom2 = vae.quant_conv(vae.encoder(g["vae_x_stage1"].to(DEV)))
        e_m2 = rel_l2(mom2, g["vae_moments2"])
    print(f"vae: denoise-encoder moments {e_m:.3e}, decoder {e_d:.3e}, encoder moments {e_m2:.3e}")
    assert e_m <= 2e-2 and e_d <= 2e-2 and e_m2 <= 2e-2
    assert rel_l2(z, g["vae_z"]) <= 2e-2


def test_vae_groupnorm_statistics_from_the_conv_epilogues():
    """The VAE with its convolutions on tiles 39 / 40 of csrc/gemm16.hip and every Normalize (model.py:48-51) taking its statistics from
    the producing epilogue (4-channel unit partials -> supir_groupnorm_parts_finalize -> `given`), at 256 px (every feature map a whole
    number of 256-pixel tile rows).  Checked: the producer path is really taken; every producer statistic against fp64 torch over the
    tensor (rstd to 1e-6 -- the statistics pass it replaces is no closer); and end to end.  The random-weight VAE amplifies a 3e-8
    perturbation of one statistic to ~1e-2 at the output (profiles/r04/diag_vae_groupnorm_statistics_1024px.log), so the end-to-end bar
    is relative: the producer path must sit as close to a run with fp64-exact statistics as the statistics-pass path does."""
    from supir_amd import _lib, ops
    vae = build_vae(DEV)
    z = T("z_gn", (1, 4, 32, 32))
    img = T("img_gn", (1, 3, 256, 256), scale=0.5)
    saved_tune, saved_flag, real_gn = dict(ops._TUNE), ops.USE_GN_PARTS, ops.groupnorm
    mode, errs = ["plain"], []

    def truth(x):
        Bx, C = x.shape[0], x.shape[-1]
        gx = x.double().reshape(Bx, -1, 32, C // 32)
        mean = gx.mean(dim=(1, 3))
        return mean, ((gx * gx).mean(dim=(1, 3)) - mean * mean).clamp_min(0)

    def gn(x, gamma, beta, eps, **kw):
        part = kw.get("part")
        if mode[0] == "truth":
            kw = {k: v for k, v in kw.items() if k != "part"}
            m, v = truth(x)
            return real_gn(x, gamma, beta, eps, given=torch.stack([m, v], -1).float().contiguous(), **kw)
        if mode[0] == "check" and part is not None and part.unit == 4:
            Bx, C = x.shape[0], x.shape[-1]
            HW = x.numel() // (Bx * C)
            giv = torch.empty(Bx, 32, 2, dtype=torch.float32, device=x.device)
            lib = _lib.load(x.dtype)
            _lib.check(lib.supir_groupnorm_parts_finalize(part.buf.data_ptr(), Bx, part.nchunk, C, part.unit, HW, giv.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "finalize", lib)
            m, v = truth(x)
            r, r3 = 1.0 / torch.sqrt(v + eps), 1.0 / torch.sqrt(giv[..., 1].double() + eps)
            errs.append((((r3 - r).abs() / r).max().item(), ((giv[..., 0].double() - m).abs() / v.sqrt()).max().item()))
        return real_gn(x, gamma, beta, eps, **kw)

    def run():
        with torch.no_grad():
            return vae.decoder(vae.post_quant_conv(z)).float(), vae.quant_conv(vae.encoder(img)).float()

    try:
        ops.groupnorm = gn
        ops.USE_GN_PARTS = False
        off_x, off_m = run()
        # force the new tiles wherever they fit (the autotuner picks gemm.hip tiles for some of these small maps)
        for k in [k for k in ops._TUNE if k[0] == "conv" and len(k) == 8]:
            _, Bc, H, W, Cin, Cout, stride, up = k
            M = Bc * (4 * H * W if up else H * W // (stride * stride))
            if Cin % 64 == 0 and M % 256 == 0 and Cout % 128 == 0 and Cout % 80:
                ops._TUNE[k] = 40 if Cout % 256 == 0 else 39
        ops.USE_GN_PARTS = True
        mode[0] = "check"
        tr = ops.start_trace()
        on_x, on_m = run()
        ops.stop_trace()
        mode[0] = "truth"
        tr_x, tr_m = run()
    finally:
        ops.groupnorm = real_gn
        ops._TUNE.clear()
        ops._TUNE.update(saved_tune)
        ops.USE_GN_PARTS = saved_flag
    n_fin = sum(1 for r in tr if r["kernel"] == "groupnorm_parts_finalize")
    n_gn = sum(1 for r in tr if r["kernel"] == "groupnorm")
    e_on, e_off = max(rel_l2(on_x, tr_x), rel_l2(on_m, tr_m)), max(rel_l2(off_x, tr_x), rel_l2(off_m, tr_m))
    print(f"vae with producer statistics: {n_fin} of {n_gn} GroupNorms; worst rstd error {max(e[0] for e in errs):.2e}, worst mean error "
          f"{max(e[1] for e in errs):.2e} std; vs a run on fp64 statistics: producer path {e_on:.3e}, statistics passes {e_off:.3e}")
    assert n_fin >= 0.8 * n_gn and len(errs) == n_fin, (n_fin, n_gn, len(errs))
    assert max(e[0] for e in errs) <= 1e-6 and max(e[1] for e in errs) <= 1e-6
    assert e_on <= 2.0 * e_off + 2e-3 and e_on <= 5e-2


def test_full_depth_wrapper_vs_oracle_on_device():
    """Full SDXL-sized model ([1,2,10] transformer depth, 3.9 G parameters) at latent 32x32: HIP path vs the oracle run in
    fp32 on the same device (the oracle is only the checker here), with the ATen-autocast bf16 floor measured beside it."""
    wrap = build_unet(depth=(1, 2, 10), device=DEV)
    sd = _oracle_sd(wrap)
    x, lq = T("xt32", (B, 4, 32, 32)), T("lq32", (B, 4, 32, 32))
    y, ctx = T("vector", (B, 2816)), T("context", (B, 77, 2048))
    t = torch.tensor([999, 3], dtype=torch.int64, device=DEV)
    cond = {"crossattn": ctx, "vector": y, "control": lq}
    ref, floor = _bf16_floor(sd, x, t, cond)
    with torch.no_grad():
        out = wrap(x, t, cond, 1.0)
    e = rel_l2(out, ref)
    print(f"full-depth wrapper: HIP bf16 vs fp32 oracle {e:.3e}; ATen-autocast bf16 floor {floor:.3e}; eps std {ref.std().item():.3f}")
    assert e <= max(2e-2, 1.5 * floor)


def test_graph_survives_new_prompt_without_recapture(wrap):
    """A new context / vector tensor of the same shape must refresh the text K/V^T and label caches in place and replay the
    SAME captured graph (no re-capture per image)."""
    x, t, cond = _wrapper_inputs()
    cond2 = dict(cond, crossattn=T("context2", (B, 77, 2048)), vector=T("vector2", (B, 2816)))
    with torch.no_grad():
        e1 = wrap(x, t, cond, 1.0).clone()
        e2 = wrap(x, t, cond2, 1.0).clone()
        wrap.enable_graph(True)
        try:
            g1 = wrap(x, t, cond, 1.0).clone()
            n_graphs = len(wrap._graphs)
            g2 = wrap(x, t, cond2, 1.0).clone()
            g1b = wrap(x, t, cond, 1.0).clone()
            assert len(wrap._graphs) == n_graphs == 1
        finally:
            wrap.enable_graph(False)
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and torch.equal(g1b, e1)
    assert not torch.equal(e1, e2)


def test_two_stream_overlap_is_bitwise_equal_to_serial(wrap):
    x, t, cond = _wrapper_inputs()
    with torch.no_grad():
        wrap.overlap_branches = False
        a = wrap(x, t, cond, 1.0).clone()
        wrap.overlap_branches = True
        b = wrap(x, t, cond, 1.0).clone()
        c = wrap(x * 0.9, t, cond, 1.0).clone()
        wrap.overlap_branches = False
        c_ref = wrap(x * 0.9, t, cond, 1.0).clone()
        wrap.overlap_branches = True
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(c, c_ref)


def test_tiled_vae_vs_reference_golden(g):
    """Native tiled VAE (all tiles resident, layer-major, pooled GroupNorm kernels) vs the reference's VAEHook output."""
    from supir_amd.utils.tilevae import VAEHook
    vae = build_vae(DEV)
    for net in (vae.denoise_encoder, vae.decoder):
        net.original_forward = net.forward
    with torch.no_grad():
        enc = VAEHook(vae.denoise_encoder, 64, is_decoder=False)(T("img_tiled", (1, 3, 192, 160), scale=0.5))
        dec = VAEHook(vae.decoder, 8, is_decoder=True)(T("z_tiled", (1, 4, 40, 32)))
        tiny = VAEHook(vae.decoder, 64, is_decoder=True)(g["vae_z"].to(DEV) / 0.13025)   # falls through to the untiled net
    e_e, e_d = rel_l2(enc, g["tiled_enc_192x160_t64"]), rel_l2(dec, g["tiled_dec_40x32_t8"])
    print(f"tiled vae: encoder {e_e:.3e}, decoder {e_d:.3e}")
    assert tuple(enc.shape) == (1, 8, 24, 20) and tuple(dec.shape) == (1, 3, 320, 256) and tuple(tiny.shape) == (1, 3, 64, 64)
    assert e_e <= 2.5e-2 and e_d <= 2.5e-2


def test_graphs_of_two_batch_sizes_keep_their_static_buffers(wrap):
    """Alternating between two captured graphs (B=2 and B=4: single-tile and tile-batched calls of the tiled sampler) must not
    free / reallocate the cached text K/V^T or label buffers the other graph points at."""
    x, t, cond = _wrapper_inputs()
    x4, t4 = torch.cat([x, x * 0.5]), torch.cat([t, t])
    cond4 = {"crossattn": cond["crossattn"].repeat(2, 1, 1).contiguous(), "vector": cond["vector"].repeat(2, 1).contiguous(),
             "control": torch.cat([cond["control"], cond["control"] * 0.7])}
    with torch.no_grad():
        e2 = wrap(x, t, cond, 1.0).clone()
        e4 = wrap(x4, t4, cond4, 1.0).clone()
        wrap.enable_graph(True)
        try:
            outs = []
            for _ in range(3):
                outs.append((wrap(x, t, cond, 1.0).clone(), wrap(x4, t4, cond4, 1.0).clone()))
                torch.empty(64 << 20, device=DEV).fill_(1.0)   # churn the allocator between replays
        finally:
            wrap.enable_graph(False)
    for a2, a4 in outs:
        assert torch.equal(a2, e2) and torch.equal(a4, e4)
    # batch grouping may change the autotuned GEMM tile and with it the summation order of the folded-LayerNorm row statistics:
    # equal up to bf16 rounding flips amplified through the network depth, not bitwise (the reference's cuBLAS path is not
    # batch-invariant either)
    assert ((e4[:2] - e2).norm() / e2.norm()).item() < 3e-2   # two equivalent bf16 evaluations: the path's own noise floor (~1-2e-2)
