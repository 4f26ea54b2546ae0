"""The other BASELINE.json configurations, each in ITS OWN arithmetic, once on the GPU: wall seconds per image and the end-to-end fraction of
the matrix peak (algorithmic TFLOP of BASELINE.md section 2 / seconds / peak).  These are parity-test cases, not the bench line.

  config 1   512 x 512, 2 EDM steps: the ONE config BASELINE quotes in fp32 -> the fp32 service (libsupir_hip_f32.so, `--diff_dtype fp32
             --ae_dtype fp32`, peak = the 157 TFLOP/s of v_mfma_f32_16x16x4_f32) with the bf16 time beside it
  config 5   1024 x 1024, DPM++ 2M restore sampler, 8 and 4 steps, diff_dtype fp16 (options/SUPIR_v0_Juggernautv9_lightning.yaml:5) with bf16 beside
  config 3   4096 x 4096, TiledRestoreEDMSampler 128 / 64 (49 tiles) + tiled VAE (512 / 64), 50 steps, bf16

Usage: python tools/bench_configs.py [--configs 1,5,3] [--tiled-steps 50] [--tile-batch 4] [--out gpurun_out/other_configs.json]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd.configs import supir_v0_config
from supir_amd.plugin import instantiate_from_config
from supir_amd.synth import synth_param, synth_tensor

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1,5,3")
ap.add_argument("--tiled-steps", type=int, default=50)
ap.add_argument("--tile-batch", type=int, default=4)
ap.add_argument("--tiled-res", type=int, default=4096)
ap.add_argument("--out", default="gpurun_out/other_configs.json")
args = ap.parse_args()
want = {int(c) for c in args.configs.split(",")}
dev = torch.device("cuda", 0)
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.0}        # dense matrix peaks, /opt/skills/guides/MI355X_MICROARCH.md
STEP_TF = {1024: 20.281, 512: 4.763}                           # one CFG-doubled UNet + control call
VAE_TF = {1024: (4.879, 10.470), 512: (1.117, 2.515)}          # encoder / decoder


def image_tflop(P, steps):
    enc, dec = VAE_TF[P]
    return steps * STEP_TF[P] + 2 * enc + 2 * dec


def build():
    cfg = supir_v0_config(sampler="RestoreEDMSampler", sampler_device="cuda")
    with torch.device(dev):
        m = instantiate_from_config(cfg)
    with torch.no_grad():
        for k, t in m.state_dict().items():
            if t.is_floating_point() and k != "denoiser.sigmas":
                t.copy_(synth_param(k, t.shape, device=dev))
    m.model.enable_graph(True)
    return m


def cond(n=1):
    c = {"crossattn": synth_tensor("bench.c", (n, 77, 2048)).to(dev), "vector": synth_tensor("bench.v", (n, 2816)).to(dev)}
    uc = {"crossattn": synth_tensor("bench.uc", (n, 77, 2048)).to(dev), "vector": synth_tensor("bench.uv", (n, 2816)).to(dev)}
    return c, uc


def run(model, P, steps, reps=1, **kw):
    x = synth_tensor(f"img{P}", (1, 3, P, P), scale=0.5).clamp(-1, 1).to(dev)
    a = dict(cond=cond(), num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0, seed=1234,
             color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=1.0)
    a.update(kw)
    out = model.batchify_sample(x, **a)          # warm-up (graph capture, autotune)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = model.batchify_sample(x, **a)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps, bool(torch.isfinite(out).all())


def entry(s, tflop, dtype, **more):
    return dict(s_per_image=round(s, 4), images_per_s=round(1 / s, 4), dtype=dtype, algorithmic_tflop=round(tflop, 1),
                achieved_tflops=round(tflop / s, 1), peak_tflops=PEAK[dtype], frac_of_peak_end_to_end=round(tflop / s / PEAK[dtype], 4), **more)


res = {"device": torch.cuda.get_device_name(0), "note": "algorithmic TFLOP: BASELINE.md section 2 (per CFG-doubled call / VAE pass) x the calls of the config"}
m = build()
if 1 in want:
    tf = image_tflop(512, 2)
    s, ok = run(m, 512, 2, reps=3)
    res["config1_512px_2steps_bf16"] = entry(s, tf, "bf16", finite=ok)
    # BASELINE configs[0] is fp32: the reference's own arithmetic for `--diff_dtype fp32 --ae_dtype fp32` (wrappers.py:87: autocast off)
    m.model.enable_graph(False)                   # the fp32 service is eager (no fused / graph forms: ops.has_fused)
    m.model.dtype, m.ae_dtype = torch.float32, torch.float32
    s32, ok32 = run(m, 512, 2, reps=2)
    res["config1_512px_2steps_fp32_service"] = entry(s32, tf, "fp32", finite=ok32, device_mem_peak_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))
    m.model.dtype, m.ae_dtype = torch.bfloat16, torch.bfloat16
    m.model.enable_graph(True)
    print(json.dumps(res), flush=True)
if 5 in want:
    m.sampler_config["target"] = "sgm.modules.diffusionmodules.sampling.RestoreDPMPP2MSampler"
    m.sampler_config["params"]["eta"] = 1.0
    for dt, name in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        m.model.dtype = dt
        assert m.model.effective_dtype == dt
        for steps in (8, 4):
            s, ok = run(m, 1024, steps, reps=2, cfg_scale=2.0, cfg_scale_start=2.0)
            res[f"config5_1024px_dpmpp2m_{steps}steps_{name}"] = entry(s, image_tflop(1024, steps), name, finite=ok,
                                                                      noise="brownian.BrownianTreeNoiseSampler (k-diffusion / torchsde absent)")
        print(json.dumps(res), flush=True)
    m.model.dtype = torch.bfloat16
    m.sampler_config["params"].pop("eta", None)
if 3 in want:
    m.sampler_config["target"] = "sgm.modules.diffusionmodules.sampling.TiledRestoreEDMSampler"
    m.sampler_config["params"].update(tile_size=128, tile_stride=64, tile_batch=args.tile_batch)
    m.init_tile_vae(encoder_tile_size=512, decoder_tile_size=64)
    torch.cuda.reset_peak_memory_stats()
    R = args.tiled_res
    lat = R // 8
    n_tiles = (math.ceil((lat - 128) / 64) + 1) ** 2                      # _sliding_windows(lat, lat, 128, 64)
    # tiled VAE (SUPIR/utils/tilevae.py:727-728): encoder tiles of 512 px + 32 px pad each side, decoder tiles of 64 latent + 11 pad each side;
    # work priced at the 1024^2 pass scaled by tile area
    n_enc = math.ceil((R - 64) / 512) ** 2
    n_dec = math.ceil((lat - 22) / 64) ** 2
    vae_tf = 2 * n_enc * VAE_TF[1024][0] * (576 / 1024) ** 2 + 2 * n_dec * VAE_TF[1024][1] * (86 * 8 / 1024) ** 2
    tf = n_tiles * args.tiled_steps * STEP_TF[1024] + vae_tf
    x = synth_tensor(f"img{R}", (1, 3, R, R), scale=0.5).clamp(-1, 1).to(dev)
    kw = dict(cond=cond(), restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0, seed=1234, color_fix_type="Wavelet",
              use_linear_CFG=True, cfg_scale_start=1.0)
    t0 = time.time()
    out = m.batchify_sample(x, num_steps=2, **kw)                          # warm-up: autotune, graph capture
    torch.cuda.synchronize()
    t_first = time.time() - t0
    t0 = time.time()
    out = m.batchify_sample(x, num_steps=args.tiled_steps, **kw)
    torch.cuda.synchronize()
    s = time.time() - t0
    res[f"config3_{R}px_tiled_bf16"] = entry(s, tf, "bf16", edm_steps=args.tiled_steps, tile_batch=args.tile_batch, latent_tiles=n_tiles,
                                             vae_tiles_enc_dec=[n_enc, n_dec], s_warmup_call_2_steps=round(t_first, 1),
                                             finite=bool(torch.isfinite(out).all()), peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))
    print(json.dumps(res), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
json.dump(res, open(args.out, "w"), indent=1)
