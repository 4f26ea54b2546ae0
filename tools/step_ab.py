"""In-process A/B of one CFG-doubled UNet+control step at 1024^2 (latent 128^2, B = 2) under hipGraph replay.
Variants are ops-level switches (autotune candidate lists); each variant re-tunes from scratch, re-captures the graph and is
timed twice, interleaved (box-to-box spread is ~5 %, so never compare numbers from two calls).
Usage: python tools/step_ab.py [variant ...]   variants: base | g32_33 | g32_33_34 | g32_33_35 | gemm16"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
variants = sys.argv[1:] or ["base", "gemm16"]
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)


VARIANTS = {"base": set(), "noqkv": {32, 33, 34, 35}, "g32_33": {32, 33}, "g32_33_34": {32, 33, 34}, "g32_33_35": {32, 33, 35}, "gemm16": {32, 33, 34, 35}}


def configure(v):
    ops.G16_TILES = VARIANTS[v]
    ops.USE_GEMM16 = bool(VARIANTS[v])
    ops.USE_QKV = v != "noqkv"
    ops._TUNE.clear()
    ops._CHOICE.clear()


res = {}
outs = {}
with torch.no_grad():
    for rep in range(2):
        for v in variants:
            configure(v)
            wrap.enable_graph(False)
            for _ in range(2):
                o = wrap(x, t, cond, 1.0)       # eager: autotune
            wrap.enable_graph(True)
            for _ in range(3):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            n = 10
            t0 = time.time()
            for _ in range(n):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            res.setdefault(v, []).append(round(ms, 2))
            outs[v] = o.clone()
            picks = {}
            for k, tl in ops._TUNE.items():
                if k[0] == "gemm":
                    picks[str(k[1:])] = tl
            print(f"rep{rep} {v}: {ms:.2f} ms/step; tiles {json.dumps(picks)} choices {ops._CHOICE}", flush=True)
    wrap.enable_graph(False)
ref = outs[variants[0]]
for v in variants[1:]:
    print(f"{v} vs {variants[0]}: rel-L2 {((outs[v] - ref).norm() / ref.norm()).item():.3e}")
print(json.dumps(res))
