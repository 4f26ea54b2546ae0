"""Parity at PRODUCTION scale: full-depth network calls at the exact shapes of BASELINE configs 1 / 2,
`SUPIRModel.batchify_sample` end to end against `oracle.batchify_sample`, the 50-step 1024^2 run against the oracle under
ATen-bf16 autocast, and the tiled / DPM++ samplers with the real network on the GPU.

The oracle (oracle/supir_oracle.py, pinned to the real reference by tests/test_oracle_golden.py) runs in fp32 ON THE DEVICE
as the checker only; the product path is what is being measured.  Every bar has an ABSOLUTE cap next to the floor-relative
one (a floor-relative bar alone scales with whatever ATen does).  Measured errors are appended to
gpurun_out/parity_r03.json (copied to profiles/r03/parity.json and committed).

Tolerances: one bf16 network call vs the fp32 oracle: rel-L2 <= 2e-2 (absolute cap 2.5e-2), and within
1.5x of what ATen-autocast bf16 (the reference's own arithmetic, wrappers.py:87) lands from fp32 on the same inputs; HIP vs
ATen-bf16 directly <= 2.5e-2.  End-to-end runs feed errors back through the sampler, so they are graded on injected-noise
2-step runs (config 1) and, for 50 steps, against the spread between the reference's bf16 and fp32 arithmetic.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import build_unet, rel_l2, synth_tensor  # noqa: E402

DEV = "cuda"
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def record(name, **vals):
    """Append measured parity numbers to gpurun_out/parity_r03.json (best effort: the directory only exists on a gpurun box
    or a developer checkout)."""
    print(f"[parity] {name}: " + ", ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in vals.items()))
    try:
        os.makedirs(_OUT, exist_ok=True)
        path = os.path.join(_OUT, "parity_r03.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = vals
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def T(name, shape, **kw):
    return synth_tensor(name, shape, **kw).to(DEV)


def psnr(a, b, peak=2.0):
    mse = ((a.float() - b.float()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()


def _sd_of(wrap):
    sd = {}
    for pfx, mod in (("model.diffusion_model.", wrap.diffusion_model), ("model.control_model.", wrap.control_model)):
        for k, v in mod.state_dict().items():
            sd[pfx + k] = v
    return sd


@pytest.fixture(scope="module")
def model():
    """SUPIRModel exactly as bench.py builds it (supir_v0_config through the plugin), synthetic weights."""
    from supir_amd.configs import supir_v0_config
    from supir_amd.plugin import instantiate_from_config
    from supir_amd.synth import synth_param
    with torch.device(DEV):
        m = instantiate_from_config(supir_v0_config(sampler_device=DEV))
    with torch.no_grad():
        for k, t in m.state_dict().items():
            if t.is_floating_point() and k != "denoiser.sigmas":
                t.copy_(synth_param(k, t.shape, device=DEV))
    return m


@pytest.fixture(scope="module")
def full(model):
    """The full SDXL-sized UNet + control (3.9 G parameters, transformer depth [1, 2, 10]): the model's own ControlWrapper."""
    model.model.enable_graph(False)
    return model.model


# ------------------------------------------------------------------------------------------ one network call, production shapes
@pytest.mark.parametrize("lat", [64, 128])
def test_full_depth_network_call_at_production_shapes(full, lat):
    """ControlWrapper.forward, full depth, B = 2 (CFG-doubled), latent 64^2 (config 1) and 128^2 (config 2: the shapes of the
    bench) -- sgm/modules/diffusionmodules/wrappers.py:84-102."""
    from oracle import supir_oracle as O
    B = 2
    x, lq = T(f"xt{lat}", (B, 4, lat, lat)), T(f"lq{lat}", (B, 4, lat, lat))
    cond = {"crossattn": T("context", (B, 77, 2048)), "vector": T("vector", (B, 2816)), "control": lq}
    t = torch.tensor([999, 3], dtype=torch.int64, device=DEV)
    sd = _sd_of(full)
    with torch.no_grad():
        ref = O.control_wrapper(sd, x, t, cond, 1.0)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            aten = O.control_wrapper(sd, x, t, cond, 1.0).float()
        out = full(x, t, cond, 1.0)
        out2 = full(x, t, cond, 1.0)
    e, floor, e_aten = rel_l2(out, ref), rel_l2(aten, ref), rel_l2(out, aten)
    record(f"network_call_full_depth_latent{lat}", hip_vs_fp32_oracle=e, aten_bf16_vs_fp32_oracle=floor, hip_vs_aten_bf16=e_aten,
           eps_std=ref.std().item(), max_abs=(out - ref).abs().max().item())
    assert torch.isfinite(out).all() and torch.equal(out, out2)
    assert e <= 2.5e-2 and e <= max(2e-2, 1.5 * floor)
    assert e_aten <= 2.5e-2


def test_control_features_full_tensors_vs_oracle(full):
    """GLVControl.forward (SUPIR/modules/SUPIR_v0.py:499-540): all 10 feature maps, full tensors, at latent 64^2."""
    from oracle import supir_oracle as O
    B, lat = 2, 64
    x, lq = T("xt64", (B, 4, lat, lat)), T("lq64", (B, 4, lat, lat))
    ctx, y = T("context", (B, 77, 2048)), T("vector", (B, 2816))
    t = torch.tensor([999, 3], dtype=torch.int64, device=DEV)
    sd = _sd_of(full)
    with torch.no_grad():
        ref = O.glv_control(sd, lq, t, x, ctx, y, p="model.control_model.")
        hs = full.control_model(x=lq, timesteps=t, xt=x, context=ctx, y=y)
    assert len(hs) == len(ref) == 10
    errs = [rel_l2(h, r) for h, r in zip(hs, ref)]
    record("glv_control_10_maps_latent64", **{f"map{i}": e for i, e in enumerate(errs)})
    for h, r in zip(hs, ref):
        assert tuple(h.shape) == tuple(r.shape)
    assert max(errs) <= 1.5e-2, errs


# ------------------------------------------------------------------------------------------ batchify_sample end to end
def _cond(n=1):
    c = {"crossattn": T("bench.c", (n, 77, 2048)), "vector": T("bench.v", (n, 2816))}
    uc = {"crossattn": T("bench.uc", (n, 77, 2048)), "vector": T("bench.uv", (n, 2816))}
    return c, uc


def _model_sd(model):
    return {k: v for k, v in model.state_dict().items()}


@pytest.mark.parametrize("restoration_scale", [-1.0, 4.0])
def test_batchify_sample_config1_vs_oracle(model, restoration_scale):
    """BASELINE config 1 through the product's SUPIRModel.batchify_sample (SUPIR/models/SUPIR_model.py:80-136): 512^2, 2 EDM
    steps, s_churn 5, linear CFG 1 -> 4, Wavelet colour fix, every RNG draw injected; checker = oracle.batchify_sample fp32."""
    from oracle import supir_oracle as O
    P, lat, steps = 512, 64, 2
    x = T("cfg1.img", (1, 3, P, P), scale=0.5).clamp(-1, 1)
    c, uc = _cond()
    noises = {"posterior": T("cfg1.post", (1, 4, lat, lat)), "init": T("cfg1.init", (1, 4, lat, lat)),
              "steps": [T(f"cfg1.eps{i}", (1, 4, lat, lat)) for i in range(steps)]}
    model.model.enable_graph(False)
    with torch.no_grad():
        out, mid = model.batchify_sample(x, cond=(c, uc), num_steps=steps, restoration_scale=restoration_scale, s_churn=5,
                                         s_noise=1.01, cfg_scale=4.0, control_scale=1.0, seed=1234, color_fix_type="Wavelet",
                                         use_linear_CFG=True, cfg_scale_start=1.0, noises=dict(noises), return_intermediates=True)
        sd = _model_sd(model)
        ref, rmid = O.batchify_sample(sd, x, c, uc, {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v])
                                                    for k, v in noises.items()}, num_steps=steps, s_churn=5, s_noise=1.01,
                                      restoration_scale=restoration_scale, cfg_scale=4.0, cfg_scale_start=1.0,
                                      table=model.denoiser.sigmas.to(DEV))
        ref = O.wavelet_reconstruction(ref, rmid["x_stage1"])
        # the reference's own bf16 arithmetic (ATen under autocast) on the same inputs: the floor for every intermediate
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a16, amid = O.batchify_sample(sd, x, c, uc, {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v])
                                                        for k, v in noises.items()}, num_steps=steps, s_churn=5, s_noise=1.01,
                                          restoration_scale=restoration_scale, cfg_scale=4.0, cfg_scale_start=1.0,
                                          table=model.denoiser.sigmas.to(DEV))
        a16 = O.wavelet_reconstruction(a16.float(), amid["x_stage1"].float())
    errs = dict(z=rel_l2(mid["z"], rmid["z"]), x_stage1=rel_l2(mid["x_stage1"], rmid["x_stage1"]),
                z_stage1=rel_l2(mid["z_stage1"], rmid["z_stage1"]), latent=rel_l2(mid["samples"], rmid["samples"]),
                image=rel_l2(out, ref), image_psnr_db=psnr(out, ref))
    floor = dict(z=rel_l2(amid["z"], rmid["z"]), x_stage1=rel_l2(amid["x_stage1"], rmid["x_stage1"]),
                 z_stage1=rel_l2(amid["z_stage1"], rmid["z_stage1"]), latent=rel_l2(amid["samples"], rmid["samples"]),
                 image=rel_l2(a16, ref), image_psnr_db=psnr(a16, ref))
    record(f"batchify_sample_config1_512px_2steps_rcfg{restoration_scale:g}", **errs,
           **{"aten_bf16_" + k: v for k, v in floor.items()})
    assert out.shape == (1, 3, P, P) and out.dtype == torch.float32 and torch.isfinite(out).all()
    # Every stage is held to 1.5x the ATen-bf16 floor of the same quantity AND to an absolute cap.  The VAE stages compound:
    # x_stage1 = decode(z) and z_stage1 = encode(x_stage1) each see an input that already carries the previous stage's bf16
    # error, through random-init conv stacks that do not contract it (measured r02: z 8.7e-3, x_stage1 2.6e-2, z_stage1 3.0e-2).
    caps = dict(z=1.1e-2, x_stage1=3.2e-2, z_stage1=3.6e-2, latent=5e-3, image=1.5e-2)   # 1.2x the measured values (profiles/r02/parity.json)
    for k, cap in caps.items():
        assert errs[k] <= cap and errs[k] <= max(1.5 * floor[k], 0.8 * cap), (k, errs[k], floor[k])
    assert errs["image_psnr_db"] >= 40.0


def test_batchify_sample_config1_fp32_service_vs_oracle(model):
    """`--ae_dtype fp32 --diff_dtype fp32` (test.py:66-67; the reference then computes in plain fp32, SUPIR_model.py:41-69 and
    wrappers.py:87): BASELINE config 1 end to end on the fp32 service (libsupir_hip_f32.so) at FULL depth against the fp32 oracle, every
    stage.  No ATen-bf16 floor to measure against here: the bars are absolute and three orders of magnitude below the bf16 ones."""
    import warnings
    from oracle import supir_oracle as O
    P, lat, steps = 512, 64, 2
    x = T("cfg1.img", (1, 3, P, P), scale=0.5).clamp(-1, 1)
    c, uc = _cond()
    noises = {"posterior": T("cfg1.post", (1, 4, lat, lat)), "init": T("cfg1.init", (1, 4, lat, lat)),
              "steps": [T(f"cfg1.eps{i}", (1, 4, lat, lat)) for i in range(steps)]}
    model.model.enable_graph(False)
    saved = (model.ae_dtype, model.model.dtype)
    model.ae_dtype = model.model.dtype = torch.float32
    try:
        with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            out, mid = model.batchify_sample(x, cond=(c, uc), num_steps=steps, restoration_scale=4.0, s_churn=5, s_noise=1.01, cfg_scale=4.0,
                                             control_scale=1.0, seed=1234, color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=1.0,
                                             noises=dict(noises), return_intermediates=True)
        assert not [r for r in rec if issubclass(r.category, RuntimeWarning)], [str(r.message) for r in rec]
    finally:
        model.ae_dtype, model.model.dtype = saved
    def oracle():
        ref, rmid = O.batchify_sample(_model_sd(model), x, c, uc, {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v])
                                                                  for k, v in noises.items()}, num_steps=steps, s_churn=5, s_noise=1.01,
                                      restoration_scale=4.0, cfg_scale=4.0, cfg_scale_start=1.0, table=model.denoiser.sigmas.to(DEV))
        return O.wavelet_reconstruction(ref, rmid["x_stage1"]), rmid

    def errors(ref, rmid):
        return dict(z=rel_l2(mid["z"], rmid["z"]), x_stage1=rel_l2(mid["x_stage1"], rmid["x_stage1"]),
                    z_stage1=rel_l2(mid["z_stage1"], rmid["z_stage1"]), latent=rel_l2(mid["samples"], rmid["samples"]),
                    image=rel_l2(out, ref), image_psnr_db=psnr(out, ref))

    # (This test runs after the bf16 tests of this file on the same model object: it is also the regression test for element-type leaks
    # between requests -- a stale bf16 embedding schedule once served the fp32 steps their input_hint_block output: 9e-5 instead of 4e-6.)
    with torch.no_grad():
        errs = errors(*oracle())
    record("batchify_sample_config1_512px_2steps_fp32_service", **errs)
    assert out.shape == (1, 3, P, P) and out.dtype == torch.float32 and torch.isfinite(out).all()
    for k in ("z", "x_stage1", "z_stage1", "latent", "image"):
        assert errs[k] <= 2e-5, (k, errs[k])
    assert errs["image_psnr_db"] >= 90.0


def test_batchify_sample_config2_50_steps_vs_oracle_bf16(model):
    """BASELINE config 2 (1024^2, 50 EDM steps, the bench workload) with every noise injected: the product path against the
    oracle under torch.autocast(bf16) -- the reference's own arithmetic on this GPU -- and both against the
    fp32 oracle.  50 steps feed each step's error back into the next, so the bar is relative: the HIP path may not be further
    from fp32 than 1.5x what ATen-bf16 is, with absolute caps on the final latent / image."""
    from oracle import supir_oracle as O
    P, lat, steps = 1024, 128, 50
    x = T("cfg2.img", (1, 3, P, P), scale=0.5).clamp(-1, 1)
    c, uc = _cond()
    noises = {"posterior": T("cfg2.post", (1, 4, lat, lat)), "init": T("cfg2.init", (1, 4, lat, lat)),
              "steps": [T(f"cfg2.eps{i}", (1, 4, lat, lat)) for i in range(steps)]}

    def clone(n):
        return {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in n.items()}

    model.model.enable_graph(True)
    try:
        with torch.no_grad():
            out, mid = model.batchify_sample(x, cond=(c, uc), num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01,
                                             cfg_scale=4.0, control_scale=1.0, seed=1234, color_fix_type="None",
                                             use_linear_CFG=True, cfg_scale_start=1.0, noises=clone(noises),
                                             return_intermediates=True)
    finally:
        model.model.enable_graph(False)
    sd = _model_sd(model)
    table = model.denoiser.sigmas.to(DEV)
    kw = dict(num_steps=steps, s_churn=5, s_noise=1.01, restoration_scale=-1.0, cfg_scale=4.0, cfg_scale_start=1.0, table=table)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref16, m16 = O.batchify_sample(sd, x, c, uc, clone(noises), **kw)
    ref16, lat16 = ref16.float(), m16["samples"].float()
    errs = dict(latent_hip_vs_aten_bf16=rel_l2(mid["samples"], lat16), image_hip_vs_aten_bf16=rel_l2(out, ref16),
                psnr_hip_vs_aten_bf16_db=psnr(out, ref16))
    if os.environ.get("SUPIR_TEST_LONG") == "1":
        # the fp32 oracle over 50 steps at 1024^2 takes ~2 minutes of ATen fp32 on the GPU: opt-in; measured r02 (profiles/r02/
        # parity.json): latent HIP vs fp32 7.3e-4, ATen-bf16 vs fp32 9.3e-4; image PSNR 51.4 dB vs 50.0 dB
        with torch.no_grad():
            ref32, m32 = O.batchify_sample(sd, x, c, uc, clone(noises), **kw)
        errs.update(latent_hip_vs_fp32=rel_l2(mid["samples"], m32["samples"]), latent_aten_bf16_vs_fp32=rel_l2(lat16, m32["samples"]),
                    image_hip_vs_fp32=rel_l2(out, ref32), image_aten_bf16_vs_fp32=rel_l2(ref16, ref32),
                    psnr_hip_vs_fp32_db=psnr(out, ref32), psnr_aten_bf16_vs_fp32_db=psnr(ref16, ref32))
        record("batchify_sample_config2_1024px_50steps", **errs)
        assert errs["latent_hip_vs_fp32"] <= max(1.5 * errs["latent_aten_bf16_vs_fp32"], 1e-3)
        assert errs["latent_hip_vs_fp32"] <= 5e-3 and errs["psnr_hip_vs_fp32_db"] >= 40.0
    else:
        record("batchify_sample_config2_1024px_50steps_vs_aten_bf16", **errs)
    assert torch.isfinite(out).all()
    # two bf16 evaluations of the same 50-step trajectory (measured r02: latent 8.7e-4, PSNR 51.7 dB)
    assert errs["latent_hip_vs_aten_bf16"] <= 5e-3 and errs["psnr_hip_vs_aten_bf16_db"] >= 40.0


def test_num_samples_4_vs_four_single_image_runs(model):
    """test.py --num_samples N (test.py:96-101; SUPIR_model.py:90-94: the image is repeated N times and ONE batchify_sample call samples
    N variations, B = 2N in the network) as a measured product mode (VERDICT r04 missing 5): N = 4 at 1024^2 through the fused sampler
    step, the per-image embedding schedule (batch 8 tables) and hipGraph replay, every RNG draw injected, against FOUR single-image runs
    of the same path with the matching noise slices.  The two differ only in the kernels the shapes pick (M = 8192 tiles vs M = 2048
    tiles: other tile sizes, same K order per element where the tiles share it) -- the bar is the one two bf16 evaluations of the same
    trajectory get (the product vs ATen-bf16 in test_batchify_sample_config2_50_steps_vs_oracle_bf16)."""
    P, lat, steps, N = 1024, 128, 10, 4
    x = T("ns4.img", (1, 3, P, P), scale=0.5).clamp(-1, 1)
    c, uc = _cond(N)
    noises = {"posterior": T("ns4.post", (N, 4, lat, lat)), "init": T("ns4.init", (N, 4, lat, lat)),
              "steps": [T(f"ns4.eps{i}", (N, 4, lat, lat)) for i in range(steps)]}
    kw = dict(num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0, control_scale=1.0, seed=1234,
              color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=1.0, return_intermediates=True)

    def sl(d, i):
        return {k: (v[i:i + 1].clone() if torch.is_tensor(v) else [t[i:i + 1].clone() for t in v]) for k, v in d.items()}

    model.model.enable_graph(True)
    try:
        with torch.no_grad():
            out4, mid4 = model.batchify_sample(x, cond=(c, uc), num_samples=N, noises={k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v])
                                                                                        for k, v in noises.items()}, **kw)
            singles = [model.batchify_sample(x, cond=(sl(c, i), sl(uc, i)), num_samples=1, noises=sl(noises, i), **kw) for i in range(N)]
    finally:
        model.model.enable_graph(False)
    assert out4.shape == (N, 3, P, P) and torch.isfinite(out4).all()
    worst_lat, worst_psnr = 0.0, float("inf")
    for i, (o1, m1) in enumerate(singles):
        worst_lat = max(worst_lat, rel_l2(mid4["samples"][i:i + 1], m1["samples"]))
        worst_psnr = min(worst_psnr, psnr(out4[i:i + 1], o1))
    record("num_samples_4_1024px_10steps_vs_single_image_runs", latent_rel_l2_worst=worst_lat, image_psnr_db_worst=worst_psnr)
    assert not torch.equal(out4[0], out4[1])                # the four samples are four different draws
    assert worst_lat <= 5e-3 and worst_psnr >= 40.0, (worst_lat, worst_psnr)


def test_batchify_sample_config2_10_steps_vs_fp32_oracle(model):
    """The bench workload's arithmetic against the FP32 oracle in the default run (the 50-step fp32 comparison
    is opt-in because it costs two minutes): 1024^2, 10 EDM steps through the hipGraph path, every noise injected.  Ten steps
    already feed each step's error through the sampler nine times; the bar is the ATen-bf16 floor on the same inputs."""
    from oracle import supir_oracle as O
    P, lat, steps = 1024, 128, 10
    x = T("cfg2.img", (1, 3, P, P), scale=0.5).clamp(-1, 1)
    c, uc = _cond()
    noises = {"posterior": T("cfg2.post", (1, 4, lat, lat)), "init": T("cfg2.init", (1, 4, lat, lat)),
              "steps": [T(f"cfg2.eps{i}", (1, 4, lat, lat)) for i in range(steps)]}

    def clone(n):
        return {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in n.items()}

    model.model.enable_graph(True)
    try:
        with torch.no_grad():
            out, mid = model.batchify_sample(x, cond=(c, uc), num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01,
                                             cfg_scale=4.0, control_scale=1.0, seed=1234, color_fix_type="Wavelet",
                                             use_linear_CFG=True, cfg_scale_start=1.0, noises=clone(noises),
                                             return_intermediates=True)
    finally:
        model.model.enable_graph(False)
    sd = _model_sd(model)
    kw = dict(num_steps=steps, s_churn=5, s_noise=1.01, restoration_scale=-1.0, cfg_scale=4.0, cfg_scale_start=1.0,
              table=model.denoiser.sigmas.to(DEV))
    with torch.no_grad():
        ref32, m32 = O.batchify_sample(sd, x, c, uc, clone(noises), **kw)
        ref32 = O.wavelet_reconstruction(ref32, m32["x_stage1"])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref16, m16 = O.batchify_sample(sd, x, c, uc, clone(noises), **kw)
        ref16 = O.wavelet_reconstruction(ref16.float(), m16["x_stage1"].float())
    errs = dict(latent_hip_vs_fp32=rel_l2(mid["samples"], m32["samples"]), latent_aten_bf16_vs_fp32=rel_l2(m16["samples"].float(), m32["samples"]),
                image_hip_vs_fp32=rel_l2(out, ref32), image_aten_bf16_vs_fp32=rel_l2(ref16, ref32),
                psnr_hip_vs_fp32_db=psnr(out, ref32), psnr_aten_bf16_vs_fp32_db=psnr(ref16, ref32))
    record("batchify_sample_config2_1024px_10steps_vs_fp32", **errs)
    assert torch.isfinite(out).all()
    assert errs["latent_hip_vs_fp32"] <= max(1.5 * errs["latent_aten_bf16_vs_fp32"], 2e-3) and errs["latent_hip_vs_fp32"] <= 1e-2
    assert errs["image_hip_vs_fp32"] <= max(1.5 * errs["image_aten_bf16_vs_fp32"], 1e-2) and errs["psnr_hip_vs_fp32_db"] >= 40.0


def test_per_block_teacher_forced_at_full_depth(full):
    """Every module of the full-depth UNet + control fed the FP32 ORACLE'S activations at its input and compared with the oracle at
    its output: an end-to-end comparison attenuates what a deep block does wrong by the 0.4 gains of the
    synthetic residual branches that follow it; here each ResBlock / SpatialTransformer stack / adapter is on its own.
    Latent 64^2 (config-1 shapes), B = 2."""
    from oracle import supir_oracle as O
    from supir_amd import weights as Wt
    B, lat = 2, 64
    x, lq = T("xt64", (B, 4, lat, lat)), T("lq64", (B, 4, lat, lat))
    ctx, y = T("context", (B, 77, 2048)), T("vector", (B, 2816))
    t = torch.tensor([999, 3], dtype=torch.int64, device=DEV)
    sd = _sd_of(full)
    cm, dm = full.control_model, full.diffusion_model
    taps = {}
    with torch.no_grad():
        control = O.glv_control(sd, lq, t, x, ctx, y, p="model.control_model.")
        O.light_glv_unet(sd, x, t, ctx, y, control, 1.0, p="model.diffusion_model.", taps=taps)

    def nhwc(v):   # what a product module receives from its predecessor: the compute dtype, channels-last memory
        return v.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    errs = {}
    with torch.no_grad(), Wt.compute_dtype(torch.bfloat16):
        emb_c, emb_u = cm._embed(t, y), dm._embed(t, y)
        # control branch: input blocks 1.. and the middle block, each from the oracle's previous feature map
        blocks = list(cm.input_blocks)[1:] + [cm.middle_block]
        for i, blk in enumerate(blocks):
            errs[f"control.block{i + 1}"] = rel_l2(blk(nhwc(control[i]), emb_c, ctx), control[i + 1])
        # UNet encoder
        ublocks = list(dm.input_blocks)[1:]
        for i, blk in enumerate(ublocks):
            errs[f"unet.enc{i + 1}"] = rel_l2(blk(nhwc(taps[f"enc{i}"]), emb_u, ctx), taps[f"enc{i + 1}"])
        errs["unet.mid"] = rel_l2(dm.middle_block(nhwc(taps[f"enc{len(ublocks)}"]), emb_u, ctx), taps["mid"])
        # decoder: adapters and output blocks in the order LightGLVUNet.forward walks them
        a_idx, c_idx = len(dm.project_modules) - 1, len(control) - 1
        errs[f"adapter{a_idx}"] = rel_l2(dm.project_modules[a_idx](nhwc(control[c_idx]), nhwc(taps["mid"]), control_scale=1.0),
                                         taps[f"adapter{a_idx}"])
        h_prev = taps[f"adapter{a_idx}"]
        a_idx -= 1
        c_idx -= 1
        n_enc = len(ublocks)
        for i, module in enumerate(dm.output_blocks):
            skip = taps[f"enc{n_enc - i}"]
            out = dm.project_modules[a_idx](nhwc(control[c_idx]), nhwc(skip), nhwc(h_prev), control_scale=1.0)
            errs[f"adapter{a_idx}"] = rel_l2(out, taps[f"adapter{a_idx}"])
            h_in = taps[f"adapter{a_idx}"]
            a_idx -= 1
            if len(module) == 3:
                errs[f"dec{i}.res"] = rel_l2(module[0](nhwc(h_in), emb_u), taps[f"res{i}"])
                errs[f"dec{i}.st"] = rel_l2(module[1](nhwc(taps[f"res{i}"]), ctx), taps[f"st{i}"])
                xa = dm.project_modules[a_idx](nhwc(control[c_idx]), nhwc(taps[f"st{i}"]), control_scale=1.0)
                errs[f"adapter{a_idx}"] = rel_l2(xa, taps[f"adapter{a_idx}"])
                errs[f"dec{i}.up"] = rel_l2(module[2](nhwc(taps[f"adapter{a_idx}"])), taps[f"out{i}"])
                a_idx -= 1
            else:
                errs[f"dec{i}"] = rel_l2(module(nhwc(h_in), emb_u, ctx), taps[f"out{i}"])
            h_prev = taps[f"out{i}"]
            c_idx -= 1
        errs["unet.out"] = rel_l2(dm._out(nhwc(h_prev)), O._conv(sd, "model.diffusion_model.out.2", torch.nn.functional.silu(
            O._gn(sd, "model.diffusion_model.out.0", h_prev, 1e-5))))
    record("per_block_teacher_forced_full_depth_latent64", **errs)
    worst = max(errs, key=errs.get)
    # a single bf16 module (GroupNorm + conv / a 10-block transformer stack with bf16 token stream): measured 2.4e-3 .. 4.7e-3
    # (profiles/r03/parity.json); anything structurally wrong in one block is >= 5e-2 on that block
    assert errs[worst] <= 8e-3, (worst, errs[worst])


def test_tiled_sampler_config3_production_scale(full):
    """BASELINE config 3's sampler at its own scale: latent 512^2 (a 4096^2 image), tiles of 128 / stride 64 =
    49 tiles, tile_batch 4 (twelve groups of four and a remainder of ONE), full depth, 2 steps.  Checker: the oracle's tiled sampler
    driving the oracle network under ATen-bf16 autocast -- the reference's own arithmetic (wrappers.py:87); the fp32 oracle network
    (49 x 2 calls of ~2.4 s) is opt-in (SUPIR_TEST_LONG=1)."""
    from oracle import supir_oracle as O
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, TiledRestoreEDMSampler
    h = w = 512
    steps = 2
    ctx, y = T("context", (2, 77, 2048)), T("vector", (2, 2816))
    lq = T("cfg3.lq", (1, 4, h, w))
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq}
    den = DiscreteDenoiserWithControl().to(DEV)
    noises = [T(f"cfg3.eps{i}", (1, 4, h, w)) for i in range(steps)]
    x0, xc = T("cfg3.x0", (1, 4, h, w)), T("cfg3.xc", (1, 4, h, w))
    from supir_amd.modules import sampling as S
    assert len(S._sliding_windows(h, w, 128, 64)) == 49
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda t_, **kw: next(it).to(t_)
    try:
        smp = TiledRestoreEDMSampler(tile_size=128, tile_stride=64, num_steps=steps, s_churn=5, s_noise=1.01, restore_cfg=4.0,
                                     guider_config=LinearCFG(1.0, 4.0), device=DEV, tile_batch=4)
        with torch.no_grad():
            out = smp(lambda i, s, cc, cs: den(full, i, s, cc, cs), x0.clone(), cond=dict(c), uc=dict(uc), x_center=xc,
                      control_scale=1.0).float()
    finally:
        torch.randn_like = orig
    sd = _sd_of(full)
    table = den.sigmas.to(DEV)

    def denoise_fn(xin, sigma, cond, cs):
        return O.discrete_denoiser_with_control(lambda a, b_, cc, s: O.control_wrapper(sd, a, b_, cc, s), table, xin, sigma, cond, cs)

    kw = dict(tile_size=128, tile_stride=64, num_steps=steps, s_churn=5, s_noise=1.01, restore_cfg=4.0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        ref16 = O.tiled_restore_edm_sample(denoise_fn, x0.clone(), c, uc, xc, noises, **kw).float()
    errs = dict(hip_vs_aten_bf16=rel_l2(out, ref16), out_std=out.std().item())
    if os.environ.get("SUPIR_TEST_LONG") == "1":
        with torch.no_grad():
            ref32 = O.tiled_restore_edm_sample(denoise_fn, x0.clone(), c, uc, xc, noises, **kw).float()
        errs.update(hip_vs_fp32=rel_l2(out, ref32), aten_bf16_vs_fp32=rel_l2(ref16, ref32))
        assert errs["hip_vs_fp32"] <= max(1.5 * errs["aten_bf16_vs_fp32"], 5e-3)
    record("tiled_sampler_config3_latent512_49tiles_tb4_2steps_full_depth", **errs)
    assert out.shape == (1, 4, h, w) and torch.isfinite(out).all()
    assert errs["hip_vs_aten_bf16"] <= 1e-2      # two bf16 evaluations of the same 2-step tiled trajectory


def test_tiled_vae_at_2048px_vs_oracle(model):
    """The tiled VAE at a size where the tile grid is real: decoder on a 256^2 latent with 64-latent tiles
    (4 x 4 tiles of 86^2 incl. padding -> 2048^2 px), encoder on a 1536 x 2048 image with 512 px tiles (3 x 4), against the
    oracle's layer-major tiled forward (pinned to the reference's VAEHook by tests/test_oracle_golden.py) in fp32."""
    from oracle import supir_oracle as O
    from supir_amd.utils.tilevae import VAEHook
    fs = model.first_stage_model
    sd = {k[len("first_stage_model."):]: v for k, v in model.state_dict().items() if k.startswith("first_stage_model.")}
    z = T("cfg3.vae.z", (1, 4, 256, 256))
    img = T("cfg3.vae.img", (1, 3, 1536, 2048), scale=0.5).clamp(-1, 1)
    with torch.no_grad():
        dec = VAEHook(fs.decoder, 64, is_decoder=True)(z).float()
        ref_dec = O.vae_tiled_forward(sd, z, "decoder.", 64, True)
        e_dec = rel_l2(dec, ref_dec)
        del ref_dec
        enc = VAEHook(fs.denoise_encoder, 512, is_decoder=False)(img).float()
        ref_enc = O.vae_tiled_forward(sd, img, "denoise_encoder.", 512, False)
        e_enc = rel_l2(enc, ref_enc)
    record("tiled_vae_2048px", decoder_16_tiles=e_dec, encoder_12_tiles=e_enc)
    assert tuple(dec.shape) == (1, 3, 2048, 2048) and tuple(enc.shape) == (1, 8, 192, 256)
    assert e_dec <= 2.5e-2 and e_enc <= 2.5e-2


def test_dpmpp2m_config5_at_1024px_full_depth_fp16(full):
    """BASELINE config 5 at the metric's size: RestoreDPMPP2MSampler (sampling.py:422-515), 4 steps, latent
    128^2, full depth, diff_dtype fp16 (the fp16 build of the kernels), scripted noise, against the same sampler class driving the
    fp32 oracle network."""
    from oracle import supir_oracle as O
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, RestoreDPMPP2MSampler
    h = w = 128
    ctx, y = T("context", (2, 77, 2048)), T("vector", (2, 2816))
    lq = T("cfg5.lq", (1, 4, h, w))
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq}
    den = DiscreteDenoiserWithControl().to(DEV)
    sd = _sd_of(full)
    x0 = T("cfg5.x0", (1, 4, h, w))

    class Scripted:
        def __init__(self, x, *a, **k):
            self.i = 0

        def __call__(self, s, sn):
            self.i += 1
            return T(f"cfg5.eps{self.i}", (1, 4, h, w))

    outs = {}
    for name, net, dt in (("fp16", full, torch.float16), ("bf16", full, torch.bfloat16),
                          ("oracle", lambda a, b_, cc, s: O.control_wrapper(sd, a, b_, cc, s), None)):
        if dt is not None:
            full.dtype = dt
        try:
            smp = RestoreDPMPP2MSampler(num_steps=4, s_noise=1.0, eta=1.0, restore_cfg=4.0, guider_config=LinearCFG(2.0, 2.0),
                                        device=DEV, noise_sampler_cls=Scripted)
            with torch.no_grad():
                outs[name] = smp(lambda i, s, cc, cs, n=net: den(n, i, s, cc, cs), x0.clone(), cond=dict(c), uc=dict(uc),
                                 control_scale=1.0).float()
        finally:
            full.dtype = torch.bfloat16
    errs = dict(fp16_vs_fp32_oracle=rel_l2(outs["fp16"], outs["oracle"]), bf16_vs_fp32_oracle=rel_l2(outs["bf16"], outs["oracle"]))
    record("dpmpp2m_config5_latent128_full_depth_4steps", **errs)
    assert torch.isfinite(outs["fp16"]).all()
    assert errs["fp16_vs_fp32_oracle"] <= 3e-3 and errs["bf16_vs_fp32_oracle"] <= 1.5e-2


# ------------------------------------------------------------------------------------------ tiled / DPM++ samplers, real network
@pytest.fixture(scope="module")
def mini():
    return build_unet(depth=(1, 1, 2), device=DEV)


def _mini_io(h, w):
    ctx, y = T("context", (2, 77, 2048)), T("vector", (2, 2816))
    lq = T("lq_tiled", (1, 4, h, w))
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq}
    return c, uc


def test_tiled_sampler_with_real_network_vs_oracle(mini):
    """TiledRestoreEDMSampler.__call__ (sgm/modules/diffusionmodules/sampling.py:600-660) with the real (reduced-depth, real
    widths) network on the GPU: latent 48x40, tiles 32 / stride 16 (6 tiles incl. ragged last rows / columns), 3 steps,
    tile_batch 1 and 4, against oracle.tiled_restore_edm_sample driving the fp32 oracle network."""
    from oracle import supir_oracle as O
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, TiledRestoreEDMSampler
    h, w, steps = 48, 40, 3
    c, uc = _mini_io(h, w)
    den = DiscreteDenoiserWithControl().to(DEV)
    noises = [T(f"tiled.gpu.eps{i}", (1, 4, h, w)) for i in range(steps)]
    x0, xc = T("tiled.gpu.x0", (1, 4, h, w)), T("tiled.gpu.xc", (1, 4, h, w))
    outs = {}
    orig = torch.randn_like
    for tb in (1, 4):
        it = iter(noises)
        torch.randn_like = lambda t_, **kw: next(it).to(t_)
        try:
            smp = TiledRestoreEDMSampler(tile_size=32, tile_stride=16, num_steps=steps, s_churn=5, s_noise=1.01, restore_cfg=4.0,
                                         guider_config=LinearCFG(1.0, 4.0), device=DEV, tile_batch=tb)
            with torch.no_grad():
                outs[tb] = smp(lambda i, s, cc, cs: den(mini, i, s, cc, cs), x0.clone(), cond=dict(c), uc=dict(uc), x_center=xc,
                               control_scale=1.0).float()
        finally:
            torch.randn_like = orig
    sd = _sd_of(mini)
    table = den.sigmas.to(DEV)

    def denoise_fn(xin, sigma, cond, cs):
        return O.discrete_denoiser_with_control(lambda a, b_, cc, s: O.control_wrapper(sd, a, b_, cc, s), table, xin, sigma, cond, cs)

    with torch.no_grad():
        ref = O.tiled_restore_edm_sample(denoise_fn, x0.clone(), c, uc, xc, noises, tile_size=32, tile_stride=16, num_steps=steps,
                                         s_churn=5, s_noise=1.01, restore_cfg=4.0).float()
    e1, e4, e14 = rel_l2(outs[1], ref), rel_l2(outs[4], ref), rel_l2(outs[4], outs[1])
    record("tiled_sampler_real_network_48x40_t32_s16_3steps", tile_batch1_vs_oracle=e1, tile_batch4_vs_oracle=e4,
           tile_batch4_vs_1=e14)
    assert e1 <= 3e-2 and e4 <= 3e-2
    assert e14 <= 2e-2   # batching tiles only regroups rows of the same GEMMs (tile choice may differ: bf16 noise floor)


def test_dpmpp2m_sampler_with_real_network_vs_oracle_network(mini):
    """RestoreDPMPP2MSampler (sampling.py:422-515, BASELINE config 5) on the GPU: the product sampler driving the HIP network
    vs the SAME sampler class driving the fp32 oracle network, scripted noise.  (The solver arithmetic itself is pinned against
    the reference class on CPU in tests/test_host_logic.py; Karras schedule / Brownian tree are third-party: parity unpinned.)"""
    from oracle import supir_oracle as O
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, RestoreDPMPP2MSampler
    h = w = 32
    c, uc = _mini_io(h, w)
    den = DiscreteDenoiserWithControl().to(DEV)
    sd = _sd_of(mini)
    x0 = T("dpm.gpu.x0", (1, 4, h, w))

    class Scripted:
        def __init__(self, x, *a, **k):
            self.i = 0

        def __call__(self, s, sn):
            self.i += 1
            return T(f"dpm.gpu.eps{self.i}", (1, 4, h, w))

    res = {}
    for steps in (8, 4):
        outs = []
        for net in (mini, lambda a, b_, cc, s: O.control_wrapper(sd, a, b_, cc, s)):
            smp = RestoreDPMPP2MSampler(num_steps=steps, s_noise=1.0, eta=1.0, restore_cfg=4.0, guider_config=LinearCFG(2.0, 2.0),
                                        device=DEV, noise_sampler_cls=Scripted)
            with torch.no_grad():
                outs.append(smp(lambda i, s, cc, cs, n=net: den(n, i, s, cc, cs), x0.clone(), cond=dict(c), uc=dict(uc),
                                control_scale=1.0).float())
        res[f"steps{steps}"] = rel_l2(outs[0], outs[1])
        assert torch.isfinite(outs[0]).all()
    record("dpmpp2m_sampler_real_network_32x32", **res)
    assert max(res.values()) <= 3e-2


def test_graph_replay_after_eager_call_with_other_prompt(mini):
    """graph A captured with prompt A; an EAGER call of the same shapes with prompt B rewrites the shared
    text-K/V^T / label buffers in place; replaying A with the same prompt-A tensor objects must refresh them first."""
    B = 2
    x = T("xt", (B, 4, 16, 16))
    t = torch.tensor([500, 37], dtype=torch.int64, device=DEV)
    condA = {"crossattn": T("context", (B, 77, 2048)), "vector": T("vector", (B, 2816)), "control": T("lq", (B, 4, 16, 16))}
    condB = dict(condA, crossattn=T("context2", (B, 77, 2048)), vector=T("vector2", (B, 2816)))
    with torch.no_grad():
        eA = mini(x, t, condA, 1.0).clone()
        mini.enable_graph(True)
        try:
            gA = mini(x, t, condA, 1.0).clone()
            mini._forward_eager(x, t, condB, 1.0)          # same shapes, other prompt, not through the graph
            gA2 = mini(x, t, condA, 1.0).clone()
        finally:
            mini.enable_graph(False)
    assert torch.equal(gA, eA) and torch.equal(gA2, eA)
