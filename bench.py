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

T_PROCESS_START = time.time()      # --rank-report: seconds from here to the first timed image (imports, RCCL init, construction, broadcast, warm-up)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level table)
HBM_PEAK_GBPS = 8000.0
UNET_STEP_TFLOP = {128: 20.281, 64: 4.763}          # algorithmic, BASELINE.md section 2 (CFG-doubled step)
IMAGE_TFLOP_1024 = 1044.8                          # 50 steps + 2 VAE enc + 2 VAE dec


BUILD_STATS = {}   # per-rank start-up figures (construction, fill, broadcast) for --rank-report


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device, rank, world, dist_on=False):
    from supir_amd.configs import supir_v0_config
    from supir_amd.plugin import instantiate_from_config
    from supir_amd.synth import synth_param
    cfg = supir_v0_config(sampler_device=str(device))
    from supir_amd.parallel import broadcast_module_, construct_replica
    replicated = world > 1 or dist_on
    t0 = time.time()
    # rank 0 constructs on the device and fills the weights; every other rank constructs on META and only allocates: all of its
    # parameters and buffers arrive by the one broadcast below (no initialiser kernels, no host RAM, on any rank)
    model = construct_replica(lambda: instantiate_from_config(cfg), device, materialize=(rank == 0 or not replicated))
    torch.cuda.synchronize()
    t_construct = time.time() - t0
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
    BUILD_STATS.update(construct_s=round(t_construct, 2), fill_s=round(t_fill, 2), meta_construction=bool(replicated and rank != 0))
    if replicated:
        # ONE weight broadcast rank0 -> all over RCCL/xGMI, coalesced into 2^28-element buckets; nothing else is communicated.  fp32
        # masters as they are (15.9 GB, once): every rank computes what a 1-GPU run computes.  skip=(): the constructor-computed buffers
        # (the denoiser's sigma table) travel too -- ranks > 0 never ran a constructor on real memory
        t0 = time.time()
        nb = broadcast_module_(model, src=0, skip=())
        torch.cuda.synchronize()
        t_bcast = time.time() - t0
        nbytes = sum(t.numel() * t.element_size() for t in sd.values())
        BUILD_STATS.update(broadcast_s=round(t_bcast, 3), broadcast_buckets=nb, broadcast_GB=round(nbytes / 1e9, 2),
                           broadcast_GBps=round(nbytes / 1e9 / max(t_bcast, 1e-9), 1))
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
    import gc
    with torch.no_grad():
        for _ in range(2):
            model.model(x, t, cond, 1.0)
        torch.cuda.synchronize()
        # the event pair of a launch also spans whatever the HOST does between recording the first event and issuing the kernel: a
        # generation-2 garbage collection there (tens of ms on a process holding the full model) once showed up as a 54 ms "launch"
        # of one GEMM (profiles/r03/bench_gc_pause_in_kernel_profile.json) -- collect now, keep the collector off while timing
        gc.collect()
        gc.disable()
        try:
            tr = ops.start_trace(timed=True)
            model.model(x, t, cond, 1.0)
            torch.cuda.synchronize()
            tr = ops.finish_timing(ops.stop_trace())
        finally:
            gc.enable()
    agg, by_shape = collections.OrderedDict(), collections.OrderedDict()
    for r in tr:
        k = r["kernel"]
        shape = None
        if k in ("gemm", "gemm_t", "conv3x3"):
            shape = f"M{r['M']} N{r['N']} K{r['K']}" + (" geglu" if r.get("act", 0) == 2 else "")
            k = ops.gemm_tile_name(r["M"], r["N"], r.get("act", 0), conv=(k == "conv3x3"), trans=(k == "gemm_t"),
                                   tile=r.get("tile", -1), group=r.get("group", 1))
        elif k == "attn":
            shape = f"B{r['B']} H{r['H']} Tq{r['Tq']} Tk{r['Tk']}"
        elif k == "groupnorm":
            shape = f"B{r['B']} HW{r['HW']} C{r['C']}"
        elif k == "xattn_q":
            shape = f"B{r['B']} H{r['H']} T{r['T']} Tk{r['Tk']} C{r['C']}"
        if r["kernel"] == "groupnorm" and r.get("parts"):
            k = "groupnorm(statistics from the producer)"   # one launch; the others are a statistics + an apply launch
        if r.get("group", 1) == 2:
            # one grouped launch = the same layer of GLVControl and of the UNet encoder (flops / bytes in the record are both problems')
            if r["kernel"] in ("attn", "groupnorm"):
                k += " x2"
            shape = (shape or "") + " x2"
        for d, key in ((agg, k), (by_shape, (k, shape))):
            a = d.setdefault(key, dict(launches=0, us=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["us"] += r["us"]
            a["flops"] += r["flops"]
            a["bytes"] += r["bytes"]
    return agg, sum(r["us"] for r in tr), by_shape


def _pick_threads(sample, cands=(16, 32, 64, 128)):
    """Thread count for the CPU baseline, chosen on the workload itself: `sample()` (one small oracle network call) is timed at a
    few counts.  A matmul probe is misleading here -- it prefers 128-256 threads on the 2-socket host, where the oracle's mix of
    small convolutions / attention / norms runs 4-10x slower than at 32 (oversubscribed OpenMP barriers)."""
    n = os.cpu_count() or 1
    best, scan = None, {}
    for c in [c for c in cands if c <= n] or [n]:
        torch.set_num_threads(c)
        t0 = time.time()
        sample()
        dt = time.time() - t0
        scan[c] = round(dt, 2)
        if best is None or dt < best[0]:
            best = (dt, c)
        if dt > 4 * best[0]:      # far past the optimum: larger counts only get worse (and slower to try)
            break
    torch.set_num_threads(best[1])
    return best[1], best[0], scan


def _reference_network(sd):
    """(callable, description) running ONE network call through the REFERENCE's own modules (ControlWrapper(LightGLVUNet) +
    GLVControl, fp32, imported from the checkout by oracle/ref_import.py) with the weights of `sd`, or None when no checkout is
    mounted (the GPU box: /root/reference does not exist there) or its construction fails.  Construction allocates a second fp32
    copy of the 3.9 G network parameters (15.5 GB of host RAM, ~2 minutes)."""
    try:
        from oracle import ref_import as R
        if not R.available():
            return None
        ns = R.load_reference()
        net, ctl, _ = R.unet_params()
        with R.quiet():
            unet = ns.LightGLVUNet(**net).eval()
            ctrl = ns.GLVControl(**ctl).eval()
        unet.load_state_dict({k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")})
        ctrl.load_state_dict({k[len("model.control_model."):]: v for k, v in sd.items() if k.startswith("model.control_model.")})
        wrap = ns.ControlWrapper(unet, dtype=torch.float32)
        wrap.load_control_model(ctrl)
        return wrap
    except Exception as e:   # the baseline leg must never take the bench line down: fall back to the port
        log(f"[cpu_baseline] reference modules unavailable ({e!r}); timing the port")
        return None


def cpu_baseline(model, device, budget_s=60.0):
    """The CPU path timed on the host cores of this box.  kind "reference": the reference's own modules (when a checkout is mounted:
    the build container); kind "port": the oracle (fp32 PyTorch restatement of the reference path) -- always the case on the GPU
    box, where /root/reference does not exist.  profiles/r04/cpu_baseline_oracle_vs_reference_same_box.json holds both timed on one
    box on the same call (oracle/ref_vs_port_timing.py): the port is a faithful proxy.
    Sample, bounded: first ONE CFG-doubled UNet+control call at 256x256 (1.28 TFLOP, a few seconds) to get the host's rate; if
    the projection fits the budget, BASELINE config 1 END TO END (512x512, 2 EDM steps, VAE encode / decode x2 each, colour fix:
    oracle.batchify_sample, 16.8 TFLOP) -- SURVEY.md 8(d).  `value` = 1024^2 50-step images/s extrapolated by algorithmic FLOPs
    from the larger sample that ran."""
    from oracle import supir_oracle as O
    from supir_amd import ops
    from supir_amd.synth import synth_tensor
    B, latent = 2, 32
    x = synth_tensor("bench.x", (B, 4, latent, latent))
    cond = {"crossattn": synth_tensor("bench.ctx", (B, 77, 2048)), "vector": synth_tensor("bench.y", (B, 2816)),
            "control": synth_tensor("bench.lq", (B, 4, latent, latent))}
    t = torch.full((B,), 500, dtype=torch.int64)
    model.model.enable_graph(False)
    with torch.no_grad():   # algorithmic FLOPs of the small sample: counted from the HIP path's own launch trace at the same shape
        ops.start_trace()
        model.model(x.to(device), t.to(device), {k: v.to(device) for k, v in cond.items()}, 1.0)
        tflop_a = sum(r["flops"] for r in ops.stop_trace()) / 1e12
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if v.is_floating_point()}
    ref = _reference_network(sd) if os.environ.get("SUPIR_CPU_BASELINE_KIND", "auto") != "port" else None
    kind = "reference" if ref is not None else "port"

    def network(xx, tt, cc):
        """one CFG-doubled UNet + control call on the host cores, through the reference's modules or the port"""
        if ref is not None:
            return ref(xx, tt, dict(cc), 1.0)
        return O.control_wrapper(sd, xx, tt, cc, 1.0)

    def sample():
        with torch.no_grad():
            network(x, t, cond)

    threads, dt_a, scan = _pick_threads(sample)
    rate = tflop_a / dt_a
    cfg1_tflop = 2 * UNET_STEP_TFLOP[64] + 2 * 1.117 + 2 * 2.515     # 512^2 step, VAE enc / dec at 512^2
    res = {"unit": "images/s", "cores": threads, "kind": kind, "host_threads_available": os.cpu_count(),
           "thread_scan_seconds_per_256px_call": scan,
           "sample_network_call_256px": {"tflop": round(tflop_a, 3), "seconds": round(dt_a, 2), "tflops": round(rate, 3)}}
    # the metric's own unit of work on the host cores: ONE CFG-doubled UNet+control call at 1024^2 (latent 128^2,
    # B = 2, 20.28 TFLOP) through the oracle -- 50 of these are 97 % of an image
    step_1024_s = None
    if UNET_STEP_TFLOP[128] / rate <= budget_s:
        x1 = synth_tensor("bench.x", (2, 4, 128, 128))
        cond1 = dict(cond, control=synth_tensor("bench.lq", (2, 4, 128, 128)))
        # the small sample may prefer fewer threads than the production-size call (its GEMMs are 16x larger): try twice the count too
        per_threads = {}
        for th in (threads, 2 * threads):
            if th > (os.cpu_count() or 1) or (per_threads and th > 64):
                continue
            torch.set_num_threads(th)
            with torch.no_grad():
                t0 = time.time()
                network(x1, t, cond1)
                per_threads[th] = time.time() - t0
        threads = min(per_threads, key=per_threads.get)
        torch.set_num_threads(threads)
        step_1024_s = per_threads[threads]
        res["cores"] = threads
        res["unet_step_1024px_seconds_by_threads"] = {k: round(v, 2) for k, v in per_threads.items()}
        res["unet_step_1024px_cfg_doubled_s"] = round(step_1024_s, 2)
        res["unet_step_1024px_tflops"] = round(UNET_STEP_TFLOP[128] / step_1024_s, 3)
    if cfg1_tflop / rate <= budget_s:
        P, lat, steps = 512, 64, 2
        img = synth_tensor("bench.cfg1", (1, 3, P, P), scale=0.5).clamp(-1, 1)
        c = {"crossattn": synth_tensor("bench.c", (1, 77, 2048)), "vector": synth_tensor("bench.v", (1, 2816))}
        uc = {"crossattn": synth_tensor("bench.uc", (1, 77, 2048)), "vector": synth_tensor("bench.uv", (1, 2816))}
        noises = {"posterior": synth_tensor("n.p", (1, 4, lat, lat)), "init": synth_tensor("n.i", (1, 4, lat, lat)),
                  "steps": [synth_tensor(f"n.s{i}", (1, 4, lat, lat)) for i in range(steps)]}
        with torch.no_grad():
            t0 = time.time()
            out, mid = O.batchify_sample(sd, img, c, uc, noises, num_steps=steps, s_churn=5, s_noise=1.01, restoration_scale=-1.0,
                                         cfg_scale=4.0, cfg_scale_start=1.0)
            O.wavelet_reconstruction(out, mid["x_stage1"])
            dt_1 = time.time() - t0
        res["config1_end_to_end_s"] = round(dt_1, 2)
        if step_1024_s is not None:
            # 50 measured-size steps + the VAE / colour-fix tail at 1024^2 priced at config 1's measured end-to-end rate
            tail_tflop = IMAGE_TFLOP_1024 - 50 * UNET_STEP_TFLOP[128]
            res["value"] = 1.0 / (50 * step_1024_s + tail_tflop / (cfg1_tflop / dt_1))
            res["sample"] = (f"one CFG-doubled UNet+control call at 1024x1024 through {'the reference modules' if ref is not None else 'the oracle'} (fp32, {UNET_STEP_TFLOP[128]:.2f} TFLOP): "
                             f"{step_1024_s:.1f} s = {UNET_STEP_TFLOP[128] / step_1024_s:.3f} TFLOP/s, and BASELINE config 1 end to end "
                             f"(512x512, 2 EDM steps, {cfg1_tflop:.1f} TFLOP): {dt_1:.1f} s, on {threads} of {os.cpu_count()} host threads; "
                             f"value = 1 / (50 x the measured step + the {tail_tflop:.1f} TFLOP VAE / colour-fix tail at config 1's rate)")
            res["seconds_sample"] = dt_1 + step_1024_s
        else:
            res["value"] = 1.0 / (dt_1 * IMAGE_TFLOP_1024 / cfg1_tflop)
            res["sample"] = (f"BASELINE config 1 end to end through the oracle (512x512, 2 EDM steps, fp32, {cfg1_tflop:.1f} TFLOP): "
                             f"{dt_1:.1f} s = {cfg1_tflop / dt_1:.3f} TFLOP/s on {threads} of {os.cpu_count()} host threads; value = 1024x1024 "
                             f"50-step images/s extrapolated by FLOPs ({IMAGE_TFLOP_1024} TFLOP per image)")
            res["seconds_sample"] = dt_1
    else:
        res["value"] = 1.0 / (dt_a * IMAGE_TFLOP_1024 / tflop_a)
        res["sample"] = (f"1 CFG-doubled UNet+control call at 256x256 ({tflop_a:.3f} TFLOP) = {dt_a:.2f} s = {rate:.3f} TFLOP/s on "
                         f"{threads} of {os.cpu_count()} host threads (config 1 end to end would exceed the {budget_s:.0f} s budget); "
                         f"extrapolated by FLOPs to the {IMAGE_TFLOP_1024} TFLOP of one 1024x1024 50-step image")
        res["seconds_sample"] = dt_a
    return res


def vae_colorfix_profile(model, P, device):
    """Per-kernel-class breakdown of the per-image tail: VAE denoise-encode + decode + encode + decode and the wavelet colour fix
    at P x P, every launch bracketed by HIP events (eager), with achieved TFLOP/s (MFMA kernels) and GB/s (HBM-bound kernels)."""
    from supir_amd import ops
    from supir_amd.synth import synth_tensor
    from supir_amd.utils.colorfix import wavelet_reconstruction
    x = synth_tensor("bench.vae", (1, 3, P, P), scale=0.5).clamp(-1, 1).to(device)
    with torch.no_grad():
        def run():
            z = model.encode_first_stage_with_denoise(x, use_sample=False)
            x1 = model.decode_first_stage(z)
            z1 = model.encode_first_stage(x1)
            out = model.decode_first_stage(z1)
            return wavelet_reconstruction(out, x1)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        import gc
        gc.collect()
        gc.disable()
        try:
            ops.start_trace(timed=True)
            run()
            torch.cuda.synchronize()
            tr = ops.finish_timing(ops.stop_trace())
        finally:
            gc.enable()
    agg = collections.OrderedDict()
    for r in tr:
        k = r["kernel"]
        if k in ("gemm", "gemm_t", "conv3x3"):
            k = ops.gemm_tile_name(r["M"], r["N"], r.get("act", 0), conv=(k == "conv3x3"), trans=(k == "gemm_t"), tile=r.get("tile", -1))
        a = agg.setdefault(k, dict(launches=0, us=0.0, flops=0.0, bytes=0.0))
        a["launches"] += 1
        a["us"] += r.get("us", 0.0)
        a["flops"] += r["flops"]
        a["bytes"] += r["bytes"]
    out = {"wall_ms_eager": round(wall_ms, 2), "kernels": {}}
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        e = {"launches": v["launches"], "ms": round(v["us"] / 1e3, 3)}
        if v["flops"]:
            e["tflops"] = round(v["flops"] / v["us"] / 1e6, 1)
            e["frac_mfma_peak"] = round(v["flops"] / v["us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 3)
        else:
            e["gbps"] = round(v["bytes"] / v["us"] / 1e3, 1)
            e["frac_hbm_peak"] = round(v["bytes"] / v["us"] / 1e3 / HBM_PEAK_GBPS, 3)
        out["kernels"][k] = e
    return out


def _rocprof_kernel_name(kernel):
    """PREFIX of the name rocprofv3 prints for a gemm-family instantiation the trace names `kernel` (ops.gemm_tile_name): the leading template
    arguments up to the one that tells the instantiations of a tile apart -- later template parameters (round 6 appended the halo width) must
    not break the match (BENCH of the round's first build lost frac_graph_replay to exactly that)."""
    names = {"gemm16_kernel<128,80,2k,s3>": "gemm16_kernel<128, 80, 4, 1, 2, 3, false, false, false, 1",
             "gemm16_kernel<128,80,2k,s2>": "gemm16_kernel<128, 80, 4, 1, 2, 2, false, false, false, 1",
             "geglu_big_kernel<256,320,4x2>": "geglu_big_kernel<1, true>", "attn": "attn_d64_pipe_kernel<3, 4, true, 1, true>",
             "gemm16_kernel<256,160,1k,s3,qkv>": "gemm16_kernel<256, 160, 8, 1, 1, 3, false, false, true, 1",
             "gemm16_kernel<256,128,1k,s3,qkv>": "gemm16_kernel<256, 128, 4, 2, 1, 3, false, false, true, 1",
             "xattn_q": "xattn_q_kernel"}
    return names.get(kernel)


def replay_profile(args, tune_path):
    """THIS command again, on THIS box, inside this invocation, under `rocprofv3 --kernel-trace --stats` (one timed image of graph
    replay with exactly the kernel picks the timed region ran): per-kernel average durations under replay -- no event pairs, next-weight
    prefetch on, the other chain running beside it.  Returns {rocprof kernel name: (average us, calls)} and the CSV's path (copied to
    gpurun_out/ when that directory exists), or None when rocprofv3 is missing / the run fails (the bench line never depends on it)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    d = tempfile.mkdtemp(prefix="supir_replay_", dir="/tmp")
    cmd = [rp, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable, os.path.abspath(__file__),
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--extra-batch", "0", "--no-kernel-profile", "--no-replay-profile",
           "--tune", "file", "--tune-file", tune_path, "--edm-steps", str(args.edm_steps), "--res", str(args.res),
           "--diff-dtype", args.diff_dtype, "--save-tune", os.path.join(d, "tune_child.json")]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        subprocess.run(cmd, timeout=420, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", env=env, check=True)
        f = next(iter(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)), None)
        if f is None:
            return None
        rows = {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
        keep = f
        go = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(go):
            keep = os.path.join(go, "bench_replay_rocprofv3_kernel_stats.csv")
            shutil.copyfile(f, keep)
        return rows, os.path.relpath(keep, ROOT) if keep.startswith(ROOT) else keep
    except Exception:   # noqa: BLE001 -- a profiling extra must never take the bench line down
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


LINE_MAX_BYTES = 4096     # the driver parses the LAST stdout line out of a tail of a few KB (BENCH_r05.json: a 20 KB line -> parsed null)

_ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_graph_replay", "avg_launch_us", "avg_launch_us_graph_replay",
                  "launches_per_unet_step", "algorithmic_gflop_per_launch", "algorithmic_mb_per_launch", "share_of_step_time", "traffic",
                  "traffic_over_algorithmic")
_CPU_KEYS = ("value", "unit", "cores", "kind", "unet_step_1024px_cfg_doubled_s", "config1_end_to_end_s", "seconds_sample", "sample")
_CONFIG_KEYS = ("workload", "edm_steps", "resolution", "images_per_gpu_per_step", "parallelism", "hip_graph")


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(full, details_path=None):
    """The ONE stdout line of the bench contract, built from the full record: the contract's keys, `roofline` and `cpu_baseline` reduced to
    scalars, three supplementary scalars -- always <= LINE_MAX_BYTES (tests/test_bench_line.py).  Everything else (per-kernel breakdowns,
    per-shape counter arrays, kernel picks) is in the side file `details` names."""
    line = {k: _r(full.get(k), 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg[k] for k in _CONFIG_KEYS if k in cfg}
    rf = full.get("roofline")
    line["roofline"] = None if rf is None else {k: _r(rf.get(k)) for k in _ROOFLINE_KEYS if k in rf}
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if cb is None else {k: _r(cb.get(k), 6) for k in _CPU_KEYS if k in cb}
    for k in ("ms_per_unet_step", "ms_per_unet_step_inside_the_sampler", "unet_step_tflops", "end_to_end_tflops_per_gpu", "output_finite"):
        if full.get(k) is not None:
            line[k] = _r(full[k], 3)
    b = full.get("batched")
    if b:
        line["batched"] = {k: _r(v, 4) for k, v in b.items()}
    tail = (full.get("kernel_breakdown_vae_colorfix") or {}).get("wall_ms_eager")
    if tail is not None:
        line["vae_colorfix_tail_ms"] = tail
    if full.get("process_group"):
        line["process_group"] = full["process_group"]
    if details_path:
        line["details"] = details_path
    out = json.dumps(line)
    # the free-text fields are the only unbounded ones: shorten them until the line fits
    for obj, key in ((line.get("cpu_baseline"), "sample"), (line["config"], "workload")):
        if len(out) <= LINE_MAX_BYTES:
            break
        if obj and isinstance(obj.get(key), str):
            obj[key] = obj[key][:max(40, len(obj[key]) - (len(out) - LINE_MAX_BYTES) - 16)] + "..."
            out = json.dumps(line)
    assert len(out) <= LINE_MAX_BYTES, len(out)
    return out


def write_details(full):
    """Full record -> gpurun_out/bench_details.json (merged back by gpurun) or, without that directory, the temp dir.  Returns the path."""
    go = os.path.join(ROOT, "gpurun_out")
    try:
        try:
            os.makedirs(go, exist_ok=True)
            path = os.path.join(go, "bench_details.json")
        except OSError:
            import tempfile
            path = os.path.join(tempfile.gettempdir(), "supir_bench_details.json")
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        return None


class StdoutForTheLine:
    """fd 1 carries the ONE line of the contract and nothing else.  From construction on, whatever anything writes to stdout on its own
    -- RCCL prints a five-line version banner when its first communicator comes up (seen under torch.distributed.run) -- goes to stderr;
    inside `with`, the real stdout is back for the final print."""

    def __init__(self):
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def __enter__(self):
        sys.stdout.flush()
        os.dup2(self.real, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(2, 1)
        return False


def _fused_step_on():
    from supir_amd.modules import sampling
    return bool(sampling.FUSED_EDM_STEP)


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
    ap.add_argument("--diff-dtype", choices=["bf16", "fp16"], default="bf16",
                    help="element type of the UNet + control kernels: bf16 = BASELINE configs[1] (the metric); fp16 = the reference's "
                         "default diff_dtype, on the fp16 build of the kernels (reported in `dtype`, never the headline line)")
    ap.add_argument("--tune", choices=["box", "file"], default="box",
                    help="box (default): forget the shipped picks and time every kernel choice on THIS box during the warm-up image (a "
                         "pick made on another box costs up to 3 %% here: profiles/r03/step_variants_*.log); file: run the picks of "
                         "--tune-file / supir_amd/tune_gfx950.json as they are (what the test suite does: same kernels in every process)")
    ap.add_argument("--tune-file", default=None, help="picks to load with --tune file (e.g. the file a previous run saved)")
    ap.add_argument("--save-tune", default=None,
                    help="where rank 0 writes the picks the timed region ran with (default: gpurun_out/tune_used.json when that directory "
                         "exists), so that a profiler run of the same command can replay exactly these kernels (--tune file --tune-file ...)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--rank-report", action="store_true",
                    help="every rank prints one '[rank-report] {json}' line on stderr: construction / fill / broadcast seconds and GB/s, host RSS, "
                         "autotune entries changed by the sync, re-capture seconds, its own images/s (tools/scale_dryrun.sh)")
    ap.add_argument("--no-replay-profile", action="store_true",
                    help="skip the same-box sub-step that re-runs this command (1 image) under rocprofv3 --kernel-trace --stats for the "
                         "dominant kernel's average duration under graph replay (roofline.frac_graph_replay)")
    args = ap.parse_args()

    the_line = StdoutForTheLine()

    BUILD_STATS["imports_s"] = round(time.time() - T_PROCESS_START, 2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    # `dist_on`: the multi-GPU code path.  Also taken with ONE rank when the process was started by torch.distributed.run
    # (`--nproc-per-node 1`): the process group is then a one-rank RCCL group and supir_amd.parallel.FORCE_COLLECTIVES makes every
    # collective of the N > 1 path (weight broadcast, autotune sync, barriers, the max-over-ranks all-reduce) actually run through
    # RCCL -- the only way to execute that path on a one-GPU box (profiles/r04/bench_torchrun_nproc1.json)
    launched = "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("SUPIR_BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or launched
    if dist_on:   # before the first HIP call of the process: the HSA runtime reads it when it initialises
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist_on:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # "nccl" == RCCL on ROCm
        if world == 1:
            from supir_amd import parallel as _par
            _par.FORCE_COLLECTIVES = True

    from supir_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    from supir_amd import ops
    from supir_amd.synth import synth_tensor
    shipped = (dict(ops._TUNE), dict(ops._CHOICE))
    if args.tune == "box":
        ops._TUNE.clear()
        ops._CHOICE.clear()
    elif args.tune_file:
        ops._TUNE.clear()
        ops._CHOICE.clear()
        ops.load_tuning(args.tune_file)

    BUILD_STATS["process_group_and_library_s"] = round(time.time() - T_PROCESS_START - BUILD_STATS["imports_s"], 2)
    model, t_fill, t_bcast = build_model(device, rank, world, dist_on)
    if args.diff_dtype == "fp16":
        model.model.dtype = torch.float16
        assert model.model.effective_dtype == torch.float16, "fp16 requested but SUPIR_FP16_NATIVE=0"
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
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    t_w = time.perf_counter()
    for i in range(args.warmup):
        out = one_image(model, x, (c, uc), 1234 + rank, args.edm_steps)
    torch.cuda.synchronize()
    BUILD_STATS["warmup_s"] = round(time.perf_counter() - t_w, 2)      # autotune of unseen shapes + graph capture + the image(s)
    autotune_resynced = None
    if dist_on and args.warmup > 0:
        # every rank runs the kernels rank 0 picked (per-process autotune can otherwise differ at near-ties, i.e. replicas that
        # differ at the bf16 noise floor); a rank whose picks changed re-captures its graphs in one more untimed image
        from supir_amd import parallel
        changed = parallel.sync_autotune(src=0)
        if changed and not args.no_graph:
            model.model.enable_graph(False)
            model.model.enable_graph(True)
        ch = torch.tensor([changed], device=device)
        dist.all_reduce(ch, op=dist.ReduceOp.MAX)
        if int(ch.item()) > 0:   # all ranks run the extra image (keeps them in step; the unchanged ones just replay)
            t_r = time.perf_counter()
            out = one_image(model, x, (c, uc), 1234 + rank, args.edm_steps)
            torch.cuda.synchronize()
            BUILD_STATS["recapture_image_s"] = round(time.perf_counter() - t_r, 2)
        autotune_resynced = int(ch.item())
        BUILD_STATS.update(autotune_entries_changed_on_this_rank=int(changed), autotune_entries_changed_max_over_ranks=autotune_resynced)
    picks = {"mode": args.tune, "tile_picks": len(ops._TUNE), "choices": len(ops._CHOICE),
             "differ_from_shipped_file": sum(1 for k, v in ops._TUNE.items() if k in shipped[0] and shipped[0][k] != v)
             + sum(1 for k, v in ops._CHOICE.items() if k in shipped[1] and shipped[1][k] != v)}
    save_to = args.save_tune or (os.path.join(ROOT, "gpurun_out", "tune_used.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    if rank == 0 and save_to and args.warmup > 0:
        ops.save_tuning(save_to)
        picks["saved_to"] = os.path.relpath(save_to, ROOT)
    replay_tune = None
    if rank == 0 and world == 1 and args.warmup > 0 and not args.no_replay_profile and not args.no_kernel_profile:
        import tempfile
        replay_tune = os.path.join(tempfile.gettempdir(), f"supir_bench_picks_{os.getpid()}.json")   # the picks the timed region runs with
        ops.save_tuning(replay_tune)
    sync()
    # what the first real SCALE run needs to explain itself: everything this rank did before its first timed image
    BUILD_STATS["start_to_first_timed_image_s"] = round(time.time() - T_PROCESS_START, 2)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_image(model, x, (c, uc), 1234 + rank + 1000 * (i + 1), args.edm_steps)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0          # this rank's own clock, before the closing barrier
    sync()
    dt = time.perf_counter() - t0
    if args.rank_report:
        # one line per rank on stderr (tools/scale_dryrun.sh collects them): start-up and steady-state figures of THIS rank
        try:
            import resource
            rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
        except Exception:   # noqa: BLE001
            rss_gb = None
        rep = dict(rank=rank, world=world, device=torch.cuda.get_device_name(device), host_peak_rss_GB=None if rss_gb is None else round(rss_gb, 2),
                   device_mem_peak_GB=round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                   images_per_s_this_rank=round(args.steps * args.images_per_gpu / dt_rank, 4), timed_s_this_rank=round(dt_rank, 3),
                   timed_s_with_barrier=round(dt, 3), **BUILD_STATS)
        log("[rank-report] " + json.dumps(rep))
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(out).all())

    # ms per UNet step (one ControlWrapper call on the CFG-doubled batch), graph replay, on this rank
    extra = {}
    if rank == 0:
        lat = P // 8
        from supir_amd.synth import synth_tensor as st
        nb = 2 * int(c["vector"].shape[0])   # CFG-doubled batch of one call: 2 x images per call
        xx = st("bench.x", (nb, 4, lat, lat)).to(device)
        cond = {"crossattn": torch.cat([uc["crossattn"], c["crossattn"]]), "vector": torch.cat([uc["vector"], c["vector"]]),
                "control": st("bench.lq", (nb, 4, lat, lat)).to(device)}
        tt = torch.full((nb,), 500, dtype=torch.int64, device=device)
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
        # the same call as a step of the sampler issues it since round 4: time / label embeddings gathered from the per-image table
        # (ControlWrapper.prepare_schedule) instead of recomputed at the head of both chains
        if hasattr(model.model, "prepare_schedule"):
            with torch.no_grad():
                model.model.prepare_schedule([500], cond["vector"])
                for _ in range(3):
                    model.model.select_step(0)
                    model.model(xx, tt, cond, 1.0)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    model.model.select_step(0)
                    model.model(xx, tt, cond, 1.0)
                e1.record()
                torch.cuda.synchronize()
                model.model.end_schedule()
            extra["ms_per_unet_step_inside_the_sampler"] = e0.elapsed_time(e1) / 10
        extra["unet_step_tflops"] = UNET_STEP_TFLOP.get(lat, 0) * (nb // 2) / (ms_unet * 1e-3) if lat in UNET_STEP_TFLOP else None

    roofline, breakdown = None, None
    if rank == 0 and not args.no_kernel_profile:
        agg, total_us, by_shape = kernel_profile(model, P // 8, device)
        breakdown = {k: {"launches": v["launches"], "ms": round(v["us"] / 1e3, 3),
                         "tflops": round(v["flops"] / v["us"] / 1e6, 1) if v["flops"] else None,
                         "gbps": round(v["bytes"] / v["us"] / 1e3, 1),
                         "frac_of_peak": round(v["flops"] / v["us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 3) if v["flops"]
                         else round(v["bytes"] / v["us"] / 1e3 / HBM_PEAK_GBPS, 3)} for k, v in agg.items()}
        # the dominant kernel = the kernel (instantiation) with the largest share of the step; achieved = its algorithmic FLOPs (bytes)
        # over its launch time, i.e. the average over its launches -- the figure rocprofv3's per-kernel average duration can be held
        # against.  One instantiation serves shapes of very different work per launch (128x80 tile: K = 1280 and K = 5120), so the
        # per-shape figures and the counter-measured traffic of each shape are listed under `shapes`.
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic = {}
        if os.path.exists(pmc):   # HBM / fabric-side bytes per launch from the committed rocprofv3 --pmc passes (per shape)
            try:
                traffic = json.load(open(pmc))
            except Exception:
                traffic = {}

        def shape_traffic(kname, shp):
            tr_e = traffic.get(kname)
            if isinstance(tr_e, list):
                return next((e for e in tr_e if shp and e.get("shape", "").replace(" geglu", "") in shp), None)
            return tr_e

        name, v = max(agg.items(), key=lambda kv: kv[1]["us"])
        shapes = []
        for (kn, shp), sv in sorted(by_shape.items(), key=lambda kv: -kv[1]["us"]):
            if kn != name:
                continue
            e = {"shape": shp, "launches": sv["launches"], "avg_launch_us": round(sv["us"] / sv["launches"], 2),
                 "share_of_step_time": round(sv["us"] / total_us, 3)}
            if sv["flops"] > 0:
                e.update(tflops=round(sv["flops"] / sv["us"] / 1e6, 1), frac=round(sv["flops"] / sv["us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                         algorithmic_gflop_per_launch=round(sv["flops"] / sv["launches"] / 1e9, 3))
            else:
                e.update(gbps=round(sv["bytes"] / sv["us"] / 1e3, 1), frac=round(sv["bytes"] / sv["us"] / 1e3 / HBM_PEAK_GBPS, 4))
            e["algorithmic_mb_per_launch"] = round(sv["bytes"] / sv["launches"] / 1e6, 3)
            e["traffic"] = shape_traffic(kn, shp)
            shapes.append(e)
        if v["flops"] > 0:
            ach = v["flops"] / v["us"] / 1e6
            roofline = {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4)}
        else:
            ach = v["bytes"] / v["us"] / 1e3
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4)}
        # `traffic` = HBM / fabric-side bytes per launch of this kernel (FETCH_SIZE x 2 + WRITE_SIZE from the committed rocprofv3 --pmc
        # passes, profiles/pmc_traffic.json, averaged over its launches of the step by shape), to be read against
        # algorithmic_mb_per_launch; None when a shape of the kernel has no counter pass
        tr_bytes, tr_alg, tr_n = 0.0, 0.0, 0
        for e in shapes:
            te = e.get("traffic")
            if te and "fetch_bytes" in te:      # shapes of this instantiation that have a counter pass (the rest are a few launches)
                tr_bytes += (te["fetch_bytes"] + te["write_bytes"]) * e["launches"]
                # the counter pass's own launch (with its residual operand) defines the algorithmic bytes the ratio is taken against
                tr_alg += te.get("algorithmic_bytes", e["algorithmic_mb_per_launch"] * 1e6) * e["launches"]
                tr_n += e["launches"]
        traffic_per_launch = tr_bytes / tr_n if tr_n else None
        roofline.update(traffic=(round(traffic_per_launch) if traffic_per_launch is not None else None),
                        traffic_unit="bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                        traffic_over_algorithmic=(round(tr_bytes / tr_alg, 3) if tr_n else None),
                        traffic_launches_covered=f"{tr_n} of {v['launches']}",
                        launches_per_unet_step=v["launches"],
                        avg_launch_us=round(v["us"] / v["launches"], 2),
                        algorithmic_gflop_per_launch=round(v["flops"] / v["launches"] / 1e9, 3),
                        algorithmic_mb_per_launch=round(v["bytes"] / v["launches"] / 1e6, 3),
                        share_of_step_time=round(v["us"] / total_us, 3), shapes=shapes)
        # the same kernel under graph REPLAY (no event pairs, next-weight prefetch on, the other chain running beside it): its average
        # duration in a rocprofv3 --kernel-trace --stats run of THIS command made now, on this box, with this run's kernel picks
        want = _rocprof_kernel_name(name)
        if want is not None and v["flops"] > 0 and world == 1 and not args.no_replay_profile and replay_tune is not None:
            rp = replay_profile(args, replay_tune)
            hit = None if rp is None else next(((us, calls) for nm, (us, calls) in rp[0].items() if want in nm), None)
            if hit is not None:
                roofline.update(frac_graph_replay=round(v["flops"] / v["launches"] / (hit[0] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                avg_launch_us_graph_replay=round(hit[0], 2), graph_replay_calls=hit[1],
                                graph_replay_source=f"{rp[1]}: rocprofv3 --kernel-trace --stats of this command (1 timed image), same box, same invocation")
        # the same figures for every kernel class that takes >= 3 % of the step (the dominant one is `roofline`)
        by_kernel = []
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
            if v["us"] < 0.03 * total_us:
                continue
            if v["flops"] > 0:
                ach = v["flops"] / v["us"] / 1e6
                e = {"kernel": k, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4)}
            else:
                ach = v["bytes"] / v["us"] / 1e3
                e = {"kernel": k, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBPS, 4)}
            e.update(launches_per_unet_step=v["launches"], avg_launch_us=round(v["us"] / v["launches"], 2),
                     share_of_step_time=round(v["us"] / total_us, 3), traffic=traffic.get(k))
            by_kernel.append(e)
        extra["roofline_by_kernel"] = by_kernel
        try:
            extra["kernel_breakdown_vae_colorfix"] = vae_colorfix_profile(model, P, device)
        except Exception as e:   # a profiling extra must never take the bench line down
            extra["kernel_breakdown_vae_colorfix"] = {"error": repr(e)}
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
        full = {
            "metric": "1024px 50-step EDM denoise images/sec", "value": n_img / dt, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.diff_dtype, "data": "synthetic",
            "config": {"workload": ("configs[1]" if world == 1 else f"configs[1] on each of {world} ranks (BASELINE configs[3] shape: "
                                    f"independent images, one per GPU)") + f": {world}xMI355X {P}x{P}, {args.edm_steps} EDM steps (RestoreEDMSampler, s_churn 5, "
                                   f"linear CFG 1.0->4.0), {args.diff_dtype} MFMA UNet+GLVControl, bf16 VAE, SUPIR-v0 config, 1 image per GPU per step, "
                                   f"random-init weights", "edm_steps": args.edm_steps, "resolution": P,
                       "images_per_gpu_per_step": ipg, "parallelism": f"dp{world} (replicated weights, no collective inside a sample)",
                       "hip_graph": not args.no_graph,
                       "two_stream_overlap": bool(model.model.overlap_branches),
                       "paired_branch_launches": bool(model.model.pair_branches),
                       "fused_sampler_step": _fused_step_on()},
            "roofline": roofline, "cpu_baseline": cpu, "autotune_entries_resynced_max_over_ranks": autotune_resynced,
            "output_finite": finite, "weight_fill_s": round(t_fill, 2), "weight_broadcast_s": round(t_bcast, 2),
            "end_to_end_tflops_per_gpu": IMAGE_TFLOP_1024 * args.steps * ipg / dt if P == 1024 and args.edm_steps == 50 else None,
            "kernel_breakdown_unet_step": breakdown, "kernel_picks": picks,
        }
        full.update(extra)
        if dist_on:
            full["process_group"] = {"backend": dist.get_backend(), "world_size": world,
                                     "collectives_forced_on_one_rank": bool(world == 1)}
        # everything measured goes to a side file; stdout carries ONE short line (the driver keeps only a few KB of the tail, and a
        # 20 KB line left round 5 without a parsed record).  Nothing is printed after it, on either stream.
        details = write_details(full)
        with the_line:
            print(compact_line(full, details), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
