"""Round-4 in-process A/B of one CFG-doubled UNet+control step at 1024^2 (latent 128^2, B = 2) under hipGraph replay.

All variants run the SAME kernel picks -- timed once on this box in the first pass -- except for the one thing they switch:
  base      the round-3 configuration: exact-erf GELU in the 256 x 320 GEGLU tile (debug knob 0 = 1)
  fastgelu  the fitted GELU (x * sigmoid(x * poly(x^2)), max |error| 2.5e-5) in that epilogue (knob 0 = 0)
  t38       fastgelu + every launch that picked the 128 x 80 tile (35 / 32: 8 waves, 104-156 KB LDS, one workgroup per CU) on tile 38
            instead (128 x 80, FOUR waves, 78 KB: two workgroups per CU, so GLVControl's and the UNet encoder's launches can co-reside)
  t38k1280  as t38, but only the K = 1280 GEMMs (the 293 fixed-cost-bound launches VERDICT r03 names)
Each variant re-captures its graph and is timed twice, interleaved (box-to-box spread is ~5 %: never compare across calls).
Usage: python tools/step_ab4.py [variant ...]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib, ops
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
variants = sys.argv[1:] or ["base", "fastgelu", "t38", "t38k1280"]
# further variants (round 4, second call): all on top of fastgelu
#   w42     the 256 x 160 tile's eight waves as 4 x 2 (64 x 80 per wave) instead of 8 x 1 (32 x 160): debug knob 1
#   gn1/gn2 GroupNorm apply with ONE / TWO row batches per workgroup (debug knob 2 = 1 / 2) instead of ~32 KB per workgroup
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.supir_debug_knob.argtypes = [ctypes.c_int, ctypes.c_int]
lib.supir_debug_knob.restype = ctypes.c_int

# pass 0: tune everything on this box (round-3 candidate lists: tile 38 not offered)
ops._TUNE.clear()
ops._CHOICE.clear()
with torch.no_grad():
    wrap.enable_graph(False)
    for _ in range(2):
        wrap(x, t, cond, 1.0)
tuned, chosen = dict(ops._TUNE), dict(ops._CHOICE)


def configure(v):
    ops._TUNE.clear()
    ops._CHOICE.clear()
    ops._TUNE.update(tuned)
    ops._CHOICE.update(chosen)
    ops.G16_TILES = {32, 33, 34, 35}
    # xattn / noxattn: the fused to_q + text cross-attention launch (csrc/xattn.hip) forced on / off for every block
    ops.USE_XATTN = "noxattn" not in v
    if v == "xattn":
        for k in list(ops._CHOICE):
            if k[0] == "xattn":
                ops._CHOICE[k] = 1
    lib.supir_debug_knob(0, 1 if v == "base" else 0)
    lib.supir_debug_knob(1, 1 if "w42" in v else 0)
    lib.supir_debug_knob(2, 1 if "gn1" in v else 2 if "gn2" in v else 0)
    lib.supir_debug_knob(4, 1 if "qkv160" in v else 0)   # qkv160: the fused q|k|v launch on its 256 x 160 tile wherever that fits (the round-3 form)
    # attn3: round-3 kernel; attn4w: four waves everywhere; attnfp32sum: the policy's kernels with the fp32 row sum of rounds 1-3
    lib.supir_debug_knob(3, 1 if "attn3" in v else 3 if "attn4w" in v else 4 if "attnfp32sum" in v else 0)
    if v in ("t38", "t38k1280"):
        ops.G16_TILES = {32, 33, 34, 35, 38}
        n = 0
        for k, tl in list(ops._TUNE.items()):
            if tl in (32, 35) and k[0] in ("gemm", "conv"):
                if v == "t38k1280" and not (k[0] == "gemm" and k[3] == 1280):
                    continue
                ops._TUNE[k] = 38
                n += 1
        print(f"  [{v}] {n} shapes moved to tile 38", flush=True)


res, outs = {}, {}
with torch.no_grad():
    for rep in range(2):
        for v in variants:
            configure(v)
            wrap.enable_graph(False)
            wrap._warm = False
            for _ in range(2):
                o = wrap(x, t, cond, 1.0)
            wrap.enable_graph(True)
            for _ in range(3):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            n = 12
            t0 = time.time()
            for _ in range(n):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            res.setdefault(v, []).append(round(ms, 3))
            outs[v] = o.clone()
            print(f"rep{rep} {v}: {ms:.3f} ms/step", flush=True)
    wrap.enable_graph(False)
for kn in range(5):
    lib.supir_debug_knob(kn, 0)
ref = outs[variants[0]]
for v in variants[1:]:
    print(f"{v} vs {variants[0]}: rel-L2 {((outs[v] - ref).norm() / ref.norm()).item():.3e}")
print(json.dumps(res))
