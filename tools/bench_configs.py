"""Run the other BASELINE.json configurations once on the GPU (parity-test cases, not the bench line): wall time per image.
  config 1: 512x512, 2 EDM steps            config 3: 4096x4096 tiled sampler (128/64) + tiled VAE (512 / 64), reduced step count
  config 5: 1024x1024 DPM++ 2M restore sampler, 8 and 4 steps (Lightning config)
Usage: python tools/bench_configs.py [--tiled-steps 4] [--tile-batch 4]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd.configs import supir_v0_config
from supir_amd.plugin import instantiate_from_config
from supir_amd.synth import synth_param, synth_tensor

ap = argparse.ArgumentParser()
ap.add_argument("--tiled-steps", type=int, default=4)
ap.add_argument("--tile-batch", type=int, default=4)
ap.add_argument("--skip-tiled", action="store_true")
ap.add_argument("--only-tiled", action="store_true")
ap.add_argument("--tiled-res", type=int, default=4096)
ap.add_argument("--tiled-single", action="store_true", help="time ONE tiled call (a warm-up at 2 steps first): for 50-step runs")
ap.add_argument("--only-config5", action="store_true", help="config 5 alone (DPM++ 2M, 8 and 4 steps)")
ap.add_argument("--fp16", action="store_true",
                help="config 5 with diff_dtype fp16 -- the reference YAML's own setting (options/SUPIR_v0_Juggernautv9_lightning.yaml:5): "
                     "the fp16 build of the kernels (libsupir_hip_f16.so) instead of bf16")
args = ap.parse_args()
dev = torch.device("cuda", 0)


def build(sampler, **extra):
    cfg = supir_v0_config(sampler=sampler, sampler_device="cuda", **extra)
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
    args_ = dict(cond=cond(), num_steps=steps, restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0, seed=1234,
                 color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=1.0)
    args_.update(kw)
    out = model.batchify_sample(x, **args_)          # warm-up (graph capture, autotune)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = model.batchify_sample(x, **args_)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps, bool(torch.isfinite(out).all()), tuple(out.shape)


res = {}
m = build("RestoreEDMSampler")
if args.only_config5:
    args.skip_tiled = True
if not args.only_tiled:
    if not args.only_config5:
        s, ok, shp = run(m, 512, 2, reps=3)
        res["config1_512px_2steps"] = {"s_per_image": s, "finite": ok, "shape": shp}
        print(res, flush=True)
    if args.fp16:
        m.model.dtype = torch.float16
        assert m.model.effective_dtype == torch.float16, "SUPIR_FP16_NATIVE=0?"
    m.sampler_config["target"] = "sgm.modules.diffusionmodules.sampling.RestoreDPMPP2MSampler"
    m.sampler_config["params"]["eta"] = 1.0
    for steps in (8, 4):
        s, ok, shp = run(m, 1024, steps, reps=2, cfg_scale=2.0, cfg_scale_start=2.0)
        res[f"config5_1024px_dpmpp2m_{steps}steps"] = {"s_per_image": s, "images_per_s": 1 / s, "finite": ok,
                                                       "diff_dtype": "fp16" if args.fp16 else "bf16"}
        print(res, flush=True)
if not args.skip_tiled:
    m.sampler_config["target"] = "sgm.modules.diffusionmodules.sampling.TiledRestoreEDMSampler"
    m.sampler_config["params"].pop("eta", None)
    m.sampler_config["params"].update(tile_size=128, tile_stride=64, tile_batch=args.tile_batch)
    m.init_tile_vae(encoder_tile_size=512, decoder_tile_size=64)
    torch.cuda.reset_peak_memory_stats()
    R = args.tiled_res
    x = synth_tensor(f"img{R}", (1, 3, R, R), scale=0.5).clamp(-1, 1).to(dev)
    kw = dict(cond=cond(), restoration_scale=-1, s_churn=5, s_noise=1.01, cfg_scale=4.0, seed=1234, color_fix_type="Wavelet",
              use_linear_CFG=True, cfg_scale_start=1.0)
    t0 = time.time()
    out = m.batchify_sample(x, num_steps=2 if args.tiled_single else args.tiled_steps, **kw)   # warm-up: autotune, graph capture
    torch.cuda.synchronize()
    t_all = time.time() - t0
    t0 = time.time()
    out = m.batchify_sample(x, num_steps=args.tiled_steps, **kw)
    torch.cuda.synchronize()
    t2 = time.time() - t0
    res[f"config3_{R}px_tiled"] = {"edm_steps": args.tiled_steps, "tile_batch": args.tile_batch, "s_first_call": t_all, "s_per_image": t2,
                                   "finite": bool(torch.isfinite(out).all()), "shape": tuple(out.shape),
                                   "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9,
                                   "note": "49 latent tiles x steps network calls + tiled VAE (64 tiles) x 4; extrapolate sampler linearly to 50 steps"}
    print(res, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/configs_fp16.json" if args.fp16 else "gpurun_out/configs.json", "w"), indent=1)
