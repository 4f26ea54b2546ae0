"""Large-M evidence (round 5): (1) per-kernel-class breakdown of ONE network call at CFG batch B = 2k (k stacked 128 x 128 latent tiles of
the tiled sampler, or `--num_samples k`), every launch bracketed by HIP events, serial streams; (2) the tiled sampler itself on a 512 x 512
latent (config 3: 49 tiles) at several tile-batch sizes, fused path vs the generic path -- per step and per tile, without the VAE.
Usage: python tools/largeM_profile.py out.json [--batches 8,16] [--tile-batches 4,7,13,25] [--steps 3] [--no-generic]"""
import argparse
import collections
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from supir_amd.configs import supir_v0_config
from supir_amd.plugin import instantiate_from_config
from supir_amd.synth import synth_param, synth_tensor

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--batches", default="8,16")
ap.add_argument("--tile-batches", default="4,7,13,25")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--no-generic", action="store_true")
ap.add_argument("--latent", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
MFMA_PEAK = 2500.0

cfg = supir_v0_config(sampler="TiledRestoreEDMSampler", sampler_device="cuda")
with torch.device(dev):
    m = instantiate_from_config(cfg)
with torch.no_grad():
    for k, t in m.state_dict().items():
        if t.is_floating_point() and k != "denoiser.sigmas":
            t.copy_(synth_param(k, t.shape, device=dev))
res = {"step_breakdown": {}, "tiled_sampler": {}}


def breakdown(B):
    x = synth_tensor("lm.x", (B, 4, 128, 128)).to(dev)
    cond = {"crossattn": synth_tensor("lm.ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("lm.y", (B, 2816)).to(dev),
            "control": synth_tensor("lm.lq", (B, 4, 128, 128)).to(dev)}
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    m.model.enable_graph(False)
    ov = m.model.overlap_branches
    m.model.overlap_branches = False
    try:
        with torch.no_grad():
            for _ in range(2):
                m.model(x, t, cond, 1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.model(x, t, cond, 1.0)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            gc.collect()
            gc.disable()
            try:
                ops.start_trace(timed=True)
                m.model(x, t, cond, 1.0)
                torch.cuda.synchronize()
                tr = ops.finish_timing(ops.stop_trace())
            finally:
                gc.enable()
    finally:
        m.model.overlap_branches = ov
    agg = collections.OrderedDict()
    for r in tr:
        k = r["kernel"]
        shape = ""
        if k in ("gemm", "gemm_t", "conv3x3"):
            shape = f"M{r['M']} N{r['N']} K{r['K']}" + (" geglu" if r.get("act", 0) == 2 else "")
            k = ops.gemm_tile_name(r["M"], r["N"], r.get("act", 0), conv=(k == "conv3x3"), trans=(k == "gemm_t"), tile=r.get("tile", -1))
        elif k == "attn":
            shape = f"B{r['B']} H{r['H']} Tq{r['Tq']} Tk{r['Tk']}"
        elif k == "xattn_q":
            shape = f"B{r['B']} H{r['H']} T{r['T']} Tk{r['Tk']} C{r['C']}"
        elif k == "groupnorm":
            shape = f"B{r['B']} HW{r['HW']} C{r['C']}"
        a = agg.setdefault((k, shape), dict(launches=0, us=0.0, flops=0.0, bytes=0.0))
        a["launches"] += 1
        a["us"] += r.get("us", 0.0)
        a["flops"] += r["flops"]
        a["bytes"] += r["bytes"]
    tot_us = sum(v["us"] for v in agg.values())
    tot_fl = sum(v["flops"] for v in agg.values())
    rows = []
    for (k, shape), v in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        e = {"kernel": k, "shape": shape, "launches": v["launches"], "ms": round(v["us"] / 1e3, 3), "share": round(v["us"] / tot_us, 4)}
        if v["flops"]:
            e["tflops"] = round(v["flops"] / v["us"] / 1e6, 1)
            e["frac_mfma_peak"] = round(v["flops"] / v["us"] / 1e6 / MFMA_PEAK, 3)
        else:
            e["gbps"] = round(v["bytes"] / v["us"] / 1e3, 1)
        rows.append(e)
    byk = collections.OrderedDict()
    for e in rows:
        b = byk.setdefault(e["kernel"], dict(ms=0.0, launches=0))
        b["ms"] += e["ms"]
        b["launches"] += e["launches"]
    return {"B": B, "wall_ms_eager_serial": round(wall, 2), "sum_kernel_ms": round(tot_us / 1e3, 2), "tflop": round(tot_fl / 1e12, 2),
            "tflops_over_sum": round(tot_fl / tot_us / 1e6, 1), "by_kernel": {k: {"ms": round(v["ms"], 2), "launches": v["launches"]} for k, v in byk.items()},
            "rows": rows[:60]}


for B in [int(v) for v in args.batches.split(",") if v]:
    try:
        res["step_breakdown"][f"B{B}"] = breakdown(B)
        r = res["step_breakdown"][f"B{B}"]
        print(f"[largeM] B={B}: eager serial {r['wall_ms_eager_serial']} ms, kernels {r['sum_kernel_ms']} ms, {r['tflops_over_sum']} TFLOP/s over kernel time", flush=True)
        for e in r["rows"][:14]:
            print("   ", e, flush=True)
    except Exception as ex:   # noqa: BLE001
        res["step_breakdown"][f"B{B}"] = {"error": repr(ex)}
        print("[largeM] breakdown failed", B, repr(ex), flush=True)
    json.dump(res, open(args.out, "w"), indent=1)

# ---- the tiled sampler on a 512 x 512 latent, no VAE
from supir_amd.modules import sampling  # noqa: E402

L = args.latent
c = {"crossattn": synth_tensor("lm.c", (1, 77, 2048)).to(dev), "vector": synth_tensor("lm.v", (1, 2816)).to(dev)}
uc = {"crossattn": synth_tensor("lm.uc", (1, 77, 2048)).to(dev), "vector": synth_tensor("lm.uv", (1, 2816)).to(dev)}
lq = synth_tensor("lm.lqbig", (1, 4, L, L)).to(dev)
xc = synth_tensor("lm.xcbig", (1, 4, L, L)).to(dev)
c["control"] = lq
uc["control"] = lq
denoiser = lambda inp, sigma, cc, cs: m.denoiser(m.model, inp, sigma, cc, cs)  # noqa: E731
m.model.enable_graph(True)
sc = m.sampler_config
for kb in [int(v) for v in args.tile_batches.split(",") if v]:
    for mode in (["fused"] if args.no_generic else ["fused", "generic"]):
        try:
            sampling.FUSED_EDM_STEP = mode == "fused"
            if mode == "fused":
                denoiser.fused = (m.denoiser, m.model)
            else:
                denoiser.__dict__.pop("fused", None)
            sc["params"].update(tile_size=128, tile_stride=64, tile_batch=kb, s_churn=5, s_noise=1.01, restore_cfg=-1)
            smp = instantiate_from_config(sc)

            def run(n):
                torch.manual_seed(1234)
                with torch.no_grad():
                    out = smp(denoiser, torch.randn(1, 4, L, L, device=dev), cond=dict(c), uc=dict(uc), num_steps=n, x_center=xc)
                torch.cuda.synchronize()
                return out

            t0 = time.time()
            run(2)
            t_first = time.time() - t0
            t0 = time.time()
            out = run(args.steps)
            dt = time.time() - t0
            ntiles = len(sampling._sliding_windows(L, L, 128, 64))
            e = {"tile_batch": kb, "mode": mode, "steps": args.steps, "s_first_call": round(t_first, 2), "ms_per_step": round(dt / args.steps * 1e3, 1),
                 "ms_per_tile_step": round(dt / args.steps / ntiles * 1e3, 2), "s_per_50_steps": round(dt / args.steps * 50, 2),
                 "tflops": round(20.281 * ntiles * args.steps / dt, 1), "finite": bool(torch.isfinite(out).all()),
                 "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
            res["tiled_sampler"][f"k{kb}_{mode}"] = e
            print("[tiled]", e, flush=True)
        except Exception as ex:   # noqa: BLE001
            res["tiled_sampler"][f"k{kb}_{mode}"] = {"error": repr(ex)}
            print("[tiled] failed", kb, mode, repr(ex), flush=True)
        json.dump(res, open(args.out, "w"), indent=1)
        m.model.enable_graph(False)      # drop the graphs of this tile batch (their pools hold the activations)
        m.model.enable_graph(True)
        torch.cuda.empty_cache()
