"""The call sequence of the reference's test.py (test.py:61-104), replayed on the GPU against this package's classes WITHOUT the
reference's file (there is no /root/reference on the GPU box; tests/test_reference_test_py.py runs the file itself verbatim wherever
a checkout is mounted): build the model from a reference-style config with the text conditioner, load a reference-keyed
checkpoint, `[.half()]`, `[init_tile_vae]`, set `ae_dtype` / `model.dtype` through the attribute protocol, `.to('cuda:0')`,
PIL image -> `PIL2Tensor` -> `batchify_denoise` -> `Tensor2PIL`, `captions = ['']` -> `batchify_sample(LQ_img, captions, ...)` with
test.py's keyword set -> `Tensor2PIL(...).save`.  Checks: a PNG of the input's size comes out, the run is reproducible bit for
bit, fp16 masters (`--loading_half_params`) change the result only at the fp16-rounding level of the weights, the tiled VAE
variant stays within the tiled-VAE tolerance of the untiled one, and fp32 requests are announced (never served silently)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_reference_test_py import DEPTH, synthetic_checkpoint, write_byte_level_clip_vocab  # noqa: E402

A_PROMPT = ("Cinematic, High Contrast, highly detailed, taken using a Canon EOS R camera, hyper detailed photo - realistic maximum "
            "detail, 32k, Color Grading, ultra HD, extreme meticulous detailing, skin pore detailing, hyper sharpness, perfect without "
            "deformations.")
N_PROMPT = ("painting, oil painting, illustration, drawing, art, sketch, oil painting, cartoon, CG Style, 3D render, unreal engine, "
            "blurring, dirty, messy, worst quality, low quality, frames, watermark, signature, jpeg artifacts, deformed, lowres, "
            "over-smooth")


def _config():
    """supir_amd.configs.supir_v0_config + the conditioner block of options/SUPIR_v0.yaml:66-106 (same targets and params)."""
    from supir_amd.configs import supir_v0_config
    cfg = supir_v0_config(transformer_depth=DEPTH)
    E = "sgm.modules.encoders.modules."
    cfg["params"]["conditioner_config"] = {"target": "sgm.modules.GeneralConditionerWithControl", "params": {"emb_models": [
        {"is_trainable": False, "input_key": "txt", "target": E + "FrozenCLIPEmbedder", "params": {"layer": "hidden", "layer_idx": 11}},
        {"is_trainable": False, "input_key": "txt", "target": E + "FrozenOpenCLIPEmbedder2",
         "params": {"arch": "ViT-bigG-14", "version": "laion2b_s39b_b160k", "freeze": True, "layer": "penultimate",
                    "always_return_pooled": True, "legacy": False}},
        {"is_trainable": False, "input_key": "original_size_as_tuple", "target": E + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "crop_coords_top_left", "target": E + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "target_size_as_tuple", "target": E + "ConcatTimestepEmbedderND", "params": {"outdim": 256}}]}}
    cfg["params"]["diffusion_dtype"] = "fp16"       # the YAML's default (options/SUPIR_v0.yaml:5); test.py overrides it below
    return cfg


def _flow(tmp_path, tag, half=False, tile_vae=False, ae_dtype=torch.bfloat16, diff_dtype=torch.bfloat16, steps=4):
    from PIL import Image
    from supir_amd.plugin import instantiate_from_config
    from supir_amd.utils.imageio import PIL2Tensor, Tensor2PIL
    model = instantiate_from_config(_config()).cpu()                                   # SUPIR/util.py:36
    res = model.load_state_dict(synthetic_checkpoint(), strict=False)                  # :38-47
    assert not res.unexpected_keys
    if half:
        model = model.half()                                                           # test.py:63-64
    if tile_vae:
        model.init_tile_vae(encoder_tile_size=128, decoder_tile_size=16)               # :65-66
    model.ae_dtype = ae_dtype                                                          # :67
    model.model.dtype = diff_dtype                                                     # :68
    model = model.to("cuda:0")                                                         # :69
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, size=(12, 10, 3), dtype=np.uint8)
    lq_pil = Image.fromarray(np.kron(base, np.ones((6, 6, 1), dtype=np.uint8)))        # 60 x 72 px
    lq, h0, w0 = PIL2Tensor(lq_pil, upsacle=1, min_size=256)                           # :80
    lq = lq.unsqueeze(0).to("cuda:0")[:, :3]
    lq512, h1, w1 = PIL2Tensor(lq_pil, upsacle=1, min_size=256, fix_resize=512)        # :84
    clean = model.batchify_denoise(lq512.unsqueeze(0).to("cuda:0")[:, :3])             # :86
    clean_pil = Tensor2PIL(clean[0], h1, w1)                                           # :87
    assert clean_pil.size == (w1, h1)
    captions = [""]                                                                    # :93
    samples = model.batchify_sample(lq, captions, num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0,
                                    control_scale=1.0, seed=1234, num_samples=1, p_p=A_PROMPT, n_p=N_PROMPT,
                                    color_fix_type="Wavelet", use_linear_CFG=True, use_linear_control_scale=False,
                                    cfg_scale_start=1.0, control_scale_start=0.0)      # :97-102
    out = tmp_path / f"{tag}_0.png"
    Tensor2PIL(samples[0], h0, w0).save(out)                                           # :104
    png = np.asarray(Image.open(out))
    assert png.shape == (h0, w0, 3) and png.std() > 1.0 and torch.isfinite(samples).all()
    return model, samples.float().cpu(), png


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    old = os.environ.get("SUPIR_CLIP_TOKENIZER")
    os.environ["SUPIR_CLIP_TOKENIZER"] = write_byte_level_clip_vocab(str(tmp_path_factory.mktemp("clip_vocab")))
    yield
    if old is None:
        del os.environ["SUPIR_CLIP_TOKENIZER"]
    else:
        os.environ["SUPIR_CLIP_TOKENIZER"] = old


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def test_testpy_call_sequence_on_the_gpu(tok, tmp_path):
    model, base, png = _flow(tmp_path, "base")
    assert model.model.effective_dtype == torch.bfloat16
    # cond and uncond text features differ (the prompts went through the tokenizer and both towers)
    _, again, png2 = _flow(tmp_path, "again")
    assert torch.equal(base, again) and np.array_equal(png, png2)                      # same seed -> same bits
    # --loading_half_params: fp16 masters, bf16 kernel copies derived from them
    mh, half, _ = _flow(tmp_path, "half", half=True)
    assert all(p.dtype == torch.float16 for p in mh.parameters() if p.is_floating_point())
    e_half = _rel(half, base)
    # --use_tile_vae: the 256-px image in 128-px encoder tiles, its 32-latent in 16-latent decoder tiles (pooled GroupNorm statistics)
    _, tiled, _ = _flow(tmp_path, "tiled", tile_vae=True)
    e_tiled = _rel(tiled, base)
    # --diff_dtype fp16 (test.py's default): the fp16 build of the kernels
    mf, f16, _ = _flow(tmp_path, "fp16", diff_dtype=torch.float16)
    assert mf.model.effective_dtype == torch.float16
    e_f16 = _rel(f16, base)
    print(f"test.py flow: half-params vs fp32 masters {e_half:.3e}; tiled VAE vs untiled {e_tiled:.3e}; fp16 vs bf16 network {e_f16:.3e}")
    assert e_half <= 5e-2 and e_tiled <= 8e-2 and e_f16 <= 5e-2
    # What the REFERENCE's own arithmetic makes of the same three mode switches, on the same model / image / prompts / step count (the
    # oracle, fp32 ATen on the GPU; every RNG draw injected so that the two runs of a pair differ in the mode only): the distance
    # between "same image, other mode" is a property of the algorithm (tile seams and pooled GroupNorm statistics; fp16-rounded
    # masters; fp16 vs bf16 activations), and the product's distance must be of the reference's size, not merely under a cap.
    from oracle import supir_oracle as O
    from supir_amd.utils.imageio import PIL2Tensor
    from PIL import Image
    dev = "cuda:0"
    rng = np.random.default_rng(7)
    basei = rng.integers(0, 256, size=(12, 10, 3), dtype=np.uint8)
    lq, _, _ = PIL2Tensor(Image.fromarray(np.kron(basei, np.ones((6, 6, 1), dtype=np.uint8))), upsacle=1, min_size=256)
    lq = lq.unsqueeze(0).to(dev)[:, :3]
    with torch.no_grad():
        sd = {k: v.float() for k, v in model.state_dict().items()}
        zc = model.encode_first_stage_with_denoise(lq, use_sample=False)
        c, uc = model.prepare_condition(zc, [""], A_PROMPT, N_PROMPT, 1)
        c = {k: v.float() for k, v in c.items() if k in ("crossattn", "vector")}
        uc = {k: v.float() for k, v in uc.items() if k in ("crossattn", "vector")}
        lat = tuple(zc.shape)
        from supir_amd.synth import synth_tensor
        noises = {"posterior": synth_tensor("flow.post", lat).to(dev), "init": synth_tensor("flow.init", lat).to(dev),
                  "steps": [synth_tensor(f"flow.eps{i}", lat).to(dev) for i in range(4)]}

        def ref_run(sd_, tile=None, autocast=None):
            ns = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in noises.items()}
            kw = dict(num_steps=4, s_churn=5, s_noise=1.01, restoration_scale=-1.0, cfg_scale=4.0, cfg_scale_start=1.0,
                      table=model.denoiser.sigmas.to(dev).float(), tile_vae=tile)
            if autocast is None:
                out, mid = O.batchify_sample(sd_, lq, c, uc, ns, **kw)
            else:
                with torch.autocast("cuda", dtype=autocast):
                    out, mid = O.batchify_sample(sd_, lq, c, uc, ns, **kw)
            return O.wavelet_reconstruction(out.float(), mid["x_stage1"].float()).cpu()

        r_base = ref_run(sd)
        d_tiled = _rel(ref_run(sd, tile=(128, 16)), r_base)
        d_half = _rel(ref_run({k: v.half().float() for k, v in sd.items()}), r_base)
        r16, rbf = ref_run(sd, autocast=torch.float16), ref_run(sd, autocast=torch.bfloat16)
        d_f16 = _rel(r16, rbf)
    print(f"reference-side (oracle) distances: half-params {d_half:.3e}; tiled VAE {d_tiled:.3e}; autocast fp16 vs bf16 {d_f16:.3e}")
    from tests.test_parity_production_gpu import record
    record("testpy_flow_mode_distances", ours=dict(half=e_half, tiled=e_tiled, fp16_vs_bf16=e_f16),
           reference=dict(half=d_half, tiled=d_tiled, fp16_vs_bf16=d_f16))
    # the product's runs are bf16 evaluations: two of them that differ by a perturbation of size d end up d + (bf16 floor of this flow)
    # apart; the floor is what the reference's own bf16 vs fp16 evaluations differ by on identical inputs (d_f16)
    assert e_tiled <= 1.5 * d_tiled + d_f16, (e_tiled, d_tiled, d_f16)
    assert e_half <= 1.5 * d_half + d_f16, (e_half, d_half, d_f16)
    assert e_f16 <= 1.5 * d_f16, (e_f16, d_f16)


def test_fp32_requests_warn_or_raise_on_the_gpu(tok, tmp_path, monkeypatch):
    """With the fp32 service switched off (SUPIR_FP32_NATIVE=0); the honoured form is tests/test_fp32_gpu.py."""
    from supir_amd import weights as Wt
    monkeypatch.setattr(Wt, "FP32_NATIVE", False)
    with pytest.warns(RuntimeWarning) as rec:
        _flow(tmp_path, "fp32", ae_dtype=torch.float32, diff_dtype=torch.float32, steps=2)
    msgs = [str(w.message) for w in rec]
    assert any("ControlWrapper.dtype" in m for m in msgs) and any("SUPIRModel.ae_dtype" in m for m in msgs)
    monkeypatch.setenv("SUPIR_STRICT_DTYPE", "1")
    with pytest.raises(RuntimeError, match="ae_dtype"):
        _flow(tmp_path, "strict", ae_dtype=torch.float32, steps=2)


def test_fp32_requests_are_honoured_on_the_gpu(tok, tmp_path):
    """`--ae_dtype fp32 --diff_dtype fp32` (test.py:66-67) through the whole test.py call sequence, with and without `--use_tile_vae`: no
    downgrade warning, the fp32 service is what runs (effective_dtype, the library the process loaded), and the result sits where a bf16 evaluation of the same flow sits relative to an fp32 one (the reference's own bf16-vs-fp32
    distance for this flow is measured in test_testpy_call_sequence_on_the_gpu's oracle runs: a few 1e-2)."""
    import warnings
    from supir_amd import _lib
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m32, f32a, png_a = _flow(tmp_path, "fp32a", ae_dtype=torch.float32, diff_dtype=torch.float32, steps=2)
        _, f32t, _ = _flow(tmp_path, "fp32t", tile_vae=True, ae_dtype=torch.float32, diff_dtype=torch.float32, steps=2)
    assert not [r for r in rec if issubclass(r.category, RuntimeWarning) and "dtype" in str(r.message)]
    assert m32.model.effective_dtype == torch.float32 and _lib._lib_f32 is not None
    _, bf, _ = _flow(tmp_path, "bf16", steps=2)
    e_bf, e_tiled = _rel(bf, f32a), _rel(f32t, f32a)
    print(f"test.py flow, 2 steps: bf16 vs fp32 service {e_bf:.3e}; fp32 tiled VAE vs untiled {e_tiled:.3e}")
    assert 1e-4 <= e_bf <= 5e-2 and e_tiled <= 8e-2
