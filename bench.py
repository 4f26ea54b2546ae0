#!/usr/bin/env python
"""bench.py -- SUPIR restoration-guided EDM sampling on MI355X: 1024x1024, 50 EDM steps, images/s (BASELINE.json).

A "step" of this benchmark is ONE IMAGE through the hot path exactly as `SUPIRModel.batchify_sample` runs it
(SUPIR/models/SUPIR_model.py:80-136; test.py defaults): denoise-encode -> decode -> encode(sample) -> 50 x
[churn noise, CFG-doubled GLVControl + LightGLVUNet call, linear CFG, Euler] -> decode -> wavelet colour fix, with
random-init SDXL + SUPIR-control weights (supir_amd.synth) and a synthetic text conditioning (crossattn [1,77,2048],
vector [1,2816]); inputs are resident in HBM before the timed region.  One process per GPU; independent images are
sharded one per rank (weak scaling), weights are broadcast once from rank 0 over RCCL, no collective inside a sample.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level table)
HBM_PEAK_GBPS = 8000.0
UNET_STEP_TFLOP = {128: 20.281, 64: 4.763}          # algorithmic, BASELINE.md section 2 (CFG-doubled step)
IMAGE_TFLOP_1024 = 1044.8                          # 50 steps + 2 VAE enc + 2 VAE dec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device, rank, world):
    from supir_amd.configs import supir_v0_config
    from supir_amd.plugin import instantiate_from_config
    from supir_amd.synth import synth_param
    cfg = supir_v0_config(sampler_device=str(device))
    with torch.device(device):
        model = instantiate_from_config(cfg)
    t0 = time.time()
    sd = model.state_dict()
    if rank == 0:
        with torch.no_grad():
            for k, t in sd.items():
                if t.is_floating_point() and k != "denoiser.sigmas":
                    t.copy_(synth_param(k, t.shape, device=device))
    torch.cuda.synchronize()
    t_fill = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        # ONE weight broadcast rank0 -> all over RCCL/xGMI, coalesced into 2^28-element buckets; nothing else is communicated
        from supir_amd.parallel import broadcast_module_
        t0 = time.time()
        broadcast_module_(model, src=0, payload_dtype=torch.bfloat16)   # 7.9 GB of bf16 instead of 15.9 GB of fp32 masters
        torch.cuda.synchronize()
        t_bcast = time.time() - t0
    return model, t_fill, t_bcast


def one_image(model, x, cond, seed, edm_steps):
    return model.batchify_sample(x, cond=cond, num_steps=edm_steps, restoration_scale=-1, s_churn=5, s_noise=1.01,
                                 cfg_scale=4.0, control_scale=1.0, seed=seed, color_fix_type="Wavelet", use_linear_CFG=True,
                                 use_linear_control_scale=False, cfg_scale_start=1.0, control_scale_start=0.0)


def kernel_profile(model, latent, device):
    """One eager CFG-doubled network call with every launch bracketed by HIP events on the launch stream."""
    from supir_amd import ops
    from supir_amd.synth import synth_tensor
    B = 2
    x = synth_tensor("bench.x", (B, 4, latent, latent)).to(device)
    cond = {"crossattn": synth_tensor("bench.ctx", (B, 77, 2048)).to(device),
            "vector": synth_tensor("bench.y", (B, 2816)).to(device),
            "control": synth_tensor("bench.lq", (B, 4, latent, latent)).to(device)}
    t = torch.full((B,), 500, dtype=torch.int64, device=device)
    model.model.enable_graph(False)
    with torch.no_grad():
        for _ in range(2):
            model.model(x, t, cond, 1.0)
        torch.cuda.synchronize()
        tr = ops.start_trace(timed=True)
        model.model(x, t, cond, 1.0)
        torch.cuda.synchronize()
        tr = ops.finish_timing(ops.stop_trace())
    agg = collections.OrderedDict()
    for r in tr:
        k = r["kernel"]
        if k in ("gemm", "gemm_t", "conv3x3"):
            k = ops.gemm_tile_name(r["M"], r["N"], r.get("act", 0), conv=(k == "conv3x3"), trans=(k == "gemm_t"),
                                   tile=r.get("tile", -1))
        a = agg.setdefault(k, dict(launches=0, us=0.0, flops=0.0, bytes=0.0))
        a["launches"] += 1
        a["us"] += r["us"]
        a["flops"] += r["flops"]
        a["bytes"] += r["bytes"]
    return agg, sum(r["us"] for r in tr)


def cpu_baseline(model, device, latent=32, max_threads=32):
    """Oracle (fp32 PyTorch restatement of the reference path, kind 'port') timed on the host cores: ONE CFG-doubled
    UNet+control call at 256x256 (latent 32), full-depth weights copied from the GPU.  Bounded sample: a few seconds of CPU
    work (a first attempt with all 256 host threads at 512x512 took 508 s -- OpenMP oversubscription -- so the thread
    count is capped and reported)."""
    from oracle import supir_oracle as O
    from supir_amd import ops
    from supir_amd.synth import synth_tensor
    threads = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(threads)
    B = 2
    x = synth_tensor("bench.x", (B, 4, latent, latent))
    cond = {"crossattn": synth_tensor("bench.ctx", (B, 77, 2048)), "vector": synth_tensor("bench.y", (B, 2816)),
            "control": synth_tensor("bench.lq", (B, 4, latent, latent))}
    t = torch.full((B,), 500, dtype=torch.int64)
    # algorithmic FLOPs of this sample: counted from the HIP path's own launch trace at the same shape
    model.model.enable_graph(False)
    with torch.no_grad():
        tr = ops.start_trace()
        model.model(x.to(device), t.to(device), {k: v.to(device) for k, v in cond.items()}, 1.0)
        tflop = sum(r["flops"] for r in ops.stop_trace()) / 1e12
    sd = {}
    for pfx, mod in (("model.diffusion_model.", model.model.diffusion_model), ("model.control_model.", model.model.control_model)):
        for k, v in mod.state_dict().items():
            sd[pfx + k] = v.detach().float().cpu()
    with torch.no_grad():
        t0 = time.time()
        O.control_wrapper(sd, x, t, cond, 1.0)
        dt = time.time() - t0
    s_per_image_1024 = dt * (IMAGE_TFLOP_1024 / tflop)  # same TFLOP/s sustained over one 1024^2 50-step image
    return {"value": 1.0 / s_per_image_1024, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 CFG-doubled UNet+control call at {latent * 8}x{latent * 8} (latent {latent}, B=2, fp32, full-depth "
                      f"weights, {tflop:.3f} TFLOP) = {dt:.2f} s = {tflop / dt:.3f} TFLOP/s on {threads} host threads "
                      f"(of {os.cpu_count()}); extrapolated by FLOPs to the {IMAGE_TFLOP_1024} TFLOP of one 1024x1024 50-step image",
            "seconds_sample": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed images per rank")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--edm-steps", type=int, default=50)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--images-per-gpu", type=int, default=1,
                    help="images batched into one batchify_sample call per rank (test.py runs 1; >1 raises M of every GEMM)")
    ap.add_argument("--extra-batch", type=int, default=4,
                    help="after the timed region also report throughput with this many images per call (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # "nccl" == RCCL on ROCm

    from supir_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    from supir_amd.synth import synth_tensor

    model, t_fill, t_bcast = build_model(device, rank, world)
    model.model.enable_graph(not args.no_graph)
    P = args.res
    def make_inputs(n):
        xs = (synth_tensor(f"bench.img{rank}.{n}", (n, 3, P, P), scale=0.5).clamp(-1, 1)).to(device)
        cc = {"crossattn": synth_tensor("bench.c", (1, 77, 2048)).to(device).repeat(n, 1, 1).contiguous(),
              "vector": synth_tensor("bench.v", (1, 2816)).to(device).repeat(n, 1).contiguous()}
        uu = {"crossattn": synth_tensor("bench.uc", (1, 77, 2048)).to(device).repeat(n, 1, 1).contiguous(),
              "vector": synth_tensor("bench.uv", (1, 2816)).to(device).repeat(n, 1).contiguous()}
        return xs, cc, uu

    ipg = args.images_per_gpu
    x, c, uc = make_inputs(ipg)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = one_image(model, x, (c, uc), 1234 + rank, args.edm_steps)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_image(model, x, (c, uc), 1234 + rank + 1000 * (i + 1), args.edm_steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(out).all())

    # ms per UNet step (one ControlWrapper call on the CFG-doubled batch), graph replay, on this rank
    extra = {}
    if rank == 0:
        lat = P // 8
        from supir_amd.synth import synth_tensor as st
        xx = st("bench.x", (2, 4, lat, lat)).to(device)
        cond = {"crossattn": torch.cat([uc["crossattn"], c["crossattn"]]), "vector": torch.cat([uc["vector"], c["vector"]]),
                "control": st("bench.lq", (2, 4, lat, lat)).to(device)}
        tt = torch.full((2,), 500, dtype=torch.int64, device=device)
        with torch.no_grad():
            for _ in range(3):
                model.model(xx, tt, cond, 1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                model.model(xx, tt, cond, 1.0)
            e1.record()
            torch.cuda.synchronize()
        ms_unet = e0.elapsed_time(e1) / 10
        extra["ms_per_unet_step"] = ms_unet
        extra["unet_step_tflops"] = UNET_STEP_TFLOP.get(lat, 0) / (ms_unet * 1e-3) if lat in UNET_STEP_TFLOP else None

    roofline, breakdown = None, None
    if rank == 0 and not args.no_kernel_profile:
        agg, total_us = kernel_profile(model, P // 8, device)
        breakdown = {k: {"launches": v["launches"], "ms": round(v["us"] / 1e3, 3),
                         "tflops": round(v["flops"] / v["us"] / 1e6, 1) if v["flops"] else None,
                         "gbps": round(v["bytes"] / v["us"] / 1e3, 1)} for k, v in agg.items()}
        dom = max(agg.items(), key=lambda kv: kv[1]["us"])
        name, v = dom
        if v["flops"] > 0:
            ach = v["flops"] / v["launches"] / (v["us"] / v["launches"] * 1e-6) / 1e12
            roofline = {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                        "launches_per_unet_step": v["launches"], "avg_launch_us": round(v["us"] / v["launches"], 2),
                        "algorithmic_gflop_per_launch": round(v["flops"] / v["launches"] / 1e9, 3),
                        "share_of_step_time": round(v["us"] / total_us, 3)}
        else:
            ach = v["bytes"] / (v["us"] * 1e-6) / 1e9
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roofline["traffic"] = json.load(open(pmc)).get(name)
            except Exception:
                pass
        model.model.enable_graph(not args.no_graph)

    if rank == 0 and world == 1 and args.extra_batch > 1 and args.extra_batch != ipg:
        # supplementary: same workload with several images per call (raises M of every GEMM from 2048.. to n*2048..)
        nb = args.extra_batch
        xb, cb, ub = make_inputs(nb)
        one_image(model, xb, (cb, ub), 4321, args.edm_steps)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        one_image(model, xb, (cb, ub), 4322, args.edm_steps)
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        extra["batched"] = {"images_per_call": nb, "images_per_s": nb / tb, "s_per_call": tb}
        del xb

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(model, device)
        except Exception as e:  # e.g. host RAM too small for the fp32 weights
            cpu = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        n_img = args.steps * world * ipg
        line = {
            "metric": "1024px 50-step EDM denoise images/sec", "value": n_img / dt, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]" if world == 1 else f"configs[1] on each of {world} ranks (BASELINE configs[3] shape: "
                                    f"independent images, one per GPU)") + f": {world}xMI355X {P}x{P}, {args.edm_steps} EDM steps (RestoreEDMSampler, s_churn 5, "
                                   f"linear CFG 1.0->4.0), bf16 MFMA UNet+GLVControl+VAE, SUPIR-v0 config, 1 image per GPU per step, "
                                   f"random-init weights", "edm_steps": args.edm_steps, "resolution": P,
                       "images_per_gpu_per_step": ipg, "parallelism": f"dp{world} (replicated weights, no collective inside a sample)",
                       "hip_graph": not args.no_graph,
                       "two_stream_overlap": bool(model.model.overlap_branches)},
            "roofline": roofline, "cpu_baseline": cpu,
            "output_finite": finite, "weight_fill_s": round(t_fill, 2), "weight_broadcast_s": round(t_bcast, 2),
            "end_to_end_tflops_per_gpu": IMAGE_TFLOP_1024 * args.steps * ipg / dt if P == 1024 and args.edm_steps == 50 else None,
            "kernel_breakdown_unet_step": breakdown,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
