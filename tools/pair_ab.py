"""In-process A/B of the grouped-launch (paired branches) path: one CFG-doubled UNet+control step at 1024^2 (latent 128^2, B = 2)
under hipGraph replay.  Variants differ in ControlWrapper.pair_branches and in which launch kinds may group (ops.PAIR_KINDS); the
single-launch autotune state is shared, each variant is timed twice, interleaved (box-to-box spread is ~5 %: never compare numbers
from two calls).  Also prints the pair-autotune picks and the per-kernel breakdown of the paired eager call.
Usage: python tools/pair_ab.py [variant ...]    variants: off | all | gemm | gemm_attn | nogn | ..."""
import collections
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
VARIANTS = {"qkv": {"qkv"}, "attn": {"attn"}, "off": None, "all": {"gemm", "conv", "qkv", "attn", "gn"}, "gemm": {"gemm"}, "gemm_conv": {"gemm", "conv"},
            "gemm_conv_qkv": {"gemm", "conv", "qkv"}, "gemm_attn": {"gemm", "conv", "qkv", "attn"}, "nogn": {"gemm", "conv", "qkv", "attn"},
            "noqkv": {"gemm", "conv", "attn", "gn"}, "noattn": {"gemm", "conv", "qkv", "gn"}, "record_only": set()}
variants = sys.argv[1:] or ["off", "all"]
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)


def configure(v):
    ops.PAIR_MIN_GAIN = 0.0
    if v.startswith("gain"):        # gain20_all: group only where the grouped launch beats the overlapped singles by 20 %
        pct, v = v[4:].split("_", 1)
        ops.PAIR_MIN_GAIN = int(pct) / 100.0
        v = "force_" + v if False else v
        for k in list(ops._TUNE):
            if k[0] == "pair":
                del ops._TUNE[k]
    if v.startswith("force_"):      # every pair that has a grouped form is grouped, whatever the timing says
        kinds = VARIANTS[v[6:]]
        for k in list(ops._TUNE):
            if k[0] == "pair":
                del ops._TUNE[k]
        ops.PAIR_FORCE = True
    else:
        kinds = VARIANTS[v]
        if getattr(ops, "PAIR_FORCE", False) or True:
            for k in list(ops._TUNE):
                if k[0] == "pair":
                    del ops._TUNE[k]
        ops.PAIR_FORCE = False
    wrap.pair_branches = kinds is not None
    ops.PAIR_KINDS = kinds or set()


res, outs = {}, {}
with torch.no_grad():
    wrap.pair_branches = False
    wrap(x, t, cond, 1.0)        # cold: serial, single-launch autotune
    for rep in range(2):
        for v in variants:
            configure(v)
            wrap.enable_graph(False)
            for _ in range(3):
                o = wrap(x, t, cond, 1.0)       # eager: (pair) autotune, then a pass recorded with the pair tiles
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
            res.setdefault(v, []).append(round(ms, 2))
            outs[v] = o.clone()
            print(f"rep{rep} {v}: {ms:.2f} ms/step   picks " + json.dumps({str(k[1:]): tl for k, tl in ops._TUNE.items() if k[0] == "pair"}), flush=True)
    wrap.enable_graph(False)
    print("pair picks:", json.dumps({str(k[1:]): v for k, v in ops._TUNE.items() if k[0] == "pair"}))
    ref = outs[variants[0]]
    for v in variants[1:]:
        print(f"{v} vs {variants[0]}: rel-L2 {((outs[v] - ref).norm() / ref.norm()).item():.3e}")
    # per-kernel breakdown, eager, every launch event-timed (serial sum; pessimistic vs the overlapped replay)
    for v in variants:
        configure(v)
        wrap(x, t, cond, 1.0)
        torch.cuda.synchronize()
        tr = ops.start_trace(timed=True)
        wrap(x, t, cond, 1.0)
        torch.cuda.synchronize()
        tr = ops.finish_timing(ops.stop_trace())
        agg = collections.OrderedDict()
        for r in tr:
            k = r["kernel"]
            if k in ("gemm", "gemm_t", "conv3x3"):
                k = ops.gemm_tile_name(r["M"], r["N"], r.get("act", 0), conv=(k == "conv3x3"), trans=(k == "gemm_t"), tile=r.get("tile", -1),
                                       group=r.get("group", 1)) + f" M{r['M']} N{r['N']} K{r['K']}"
            elif r.get("group") == 2:
                k += " x2"
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += r["us"]
            a[2] += r["flops"]
        tot = sum(a[1] for a in agg.values())
        print(f"[{v}] eager serial sum {tot / 1e3:.2f} ms over {len(tr)} launches, {sum(1 for r in tr if r.get('group') == 2)} grouped")
        if v == variants[-1]:
            for k, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
                print(f"  {k:78s} x{n:4d} {us / 1e3:7.3f} ms  {us / n:7.1f} us/launch  {fl / us / 1e6 if us else 0:7.1f} TF/s")
print(json.dumps(res))
