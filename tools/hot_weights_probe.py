"""How much of a step is the price of COLD weights?  One CFG-doubled UNet+control step at 1024^2 under hipGraph replay, (a) as it
is -- 7.7 GB of distinct weights streamed from HBM once per step -- and (b) with every derived weight layout of the same shape
aliased to ONE buffer (garbage results, same launches, same shapes): the ~150 MB of distinct shapes stay in the 256 MB Infinity
Cache, so every GEMM meets its weights warm.  (b) - (a) is the ceiling of what any weight-prefetch scheme can recover.
Usage: python tools/hot_weights_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from supir_amd import weights as Wt
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)


def timed(tag, prefetch):
    wrap.prefetch_distance = prefetch
    wrap.enable_graph(False)
    wrap._warm = False
    with torch.no_grad():
        for _ in range(2):
            wrap(x, t, cond, 1.0)
        wrap.enable_graph(True)
        for _ in range(3):
            wrap(x, t, cond, 1.0)
        torch.cuda.synchronize()
        n = 12
        t0 = time.time()
        for _ in range(n):
            wrap(x, t, cond, 1.0)
        torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    print(f"{tag}: {ms:.2f} ms/step (next-weight prefetch distance {prefetch})", flush=True)
    wrap.enable_graph(False)
    return ms


res = {}
for rep in range(2):
    res.setdefault("distinct weights, prefetch 1", []).append(timed("distinct weights", 1))
    res.setdefault("distinct weights, prefetch 0", []).append(timed("distinct weights", 0))

# alias: every derived layout of a given (shape, dtype) becomes the same storage
pool = {}
for name in ("linear_w", "conv3x3_w"):
    orig = getattr(Wt, name)

    def shared(w, _orig=orig):
        v = _orig(w)
        key = (tuple(v.shape), v.dtype)
        if key not in pool:
            pool[key] = v
        return pool[key]

    setattr(Wt, name, shared)
orig_fold = Wt.fold_layernorm


def shared_fold(w, bias, gamma, beta):
    wp, cs, bp = orig_fold(w, bias, gamma, beta)
    key = (tuple(wp.shape), wp.dtype, "fold")
    if key not in pool:
        pool[key] = wp
    return pool[key], cs, bp


Wt.fold_layernorm = shared_fold
from supir_amd.modules import base  # noqa: E402
for m in list(wrap.control_model.modules()) + list(wrap.diffusion_model.modules()):
    for attr in ("_pw", "_pb", "_pw9", "_pf", "_fold", "_qk", "_gb", "_il", "_il16", "_emb_w"):
        pr = getattr(m, attr, None)
        if isinstance(pr, base.Prep):
            pr.key = None          # re-derive through the aliased constructors
torch.cuda.empty_cache()
for rep in range(2):
    res.setdefault("aliased weights (warm), prefetch 0", []).append(timed("aliased weights", 0))
    res.setdefault("aliased weights (warm), prefetch 1", []).append(timed("aliased weights", 1))
print("distinct (shape, dtype) buffers:", len(pool), "total MB", round(sum(v.numel() * v.element_size() for v in pool.values()) / 1e6, 1))
print(res)
