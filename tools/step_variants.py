"""In-process A/B of one CFG-doubled UNet+control step at 1024^2 under hipGraph replay across host-level switches:
  file    kernels picked by the shipped tune file (supir_amd/tune_gfx950.json)
  retune  autotune state cleared and re-timed on this box
  gnv1    GroupNorm apply kernel v1 (LDS table) instead of v2 (SUPIR_GN_APPLY=v1)
  nomlp   (upper bound, wrong results) the adapters' mlp_shared convolutions (Cc -> 128, small-N implicit GEMMs on the side stream) skipped:
          what making them free would be worth
  noside  (upper bound, wrong results) all adapter control sides skipped (cached outputs of an earlier call)
Each variant re-captures its graph and is timed twice, interleaved.  Usage: python tools/step_variants.py [variant ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
variants = sys.argv[1:] or ["file", "gnv1", "retune"]
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
file_tune, file_choice = dict(ops._TUNE), dict(ops._CHOICE)
retuned = None


from supir_amd.modules import supir_v0 as _S  # noqa: E402
_orig_sft_side, _orig_xattn_side = _S.ZeroSFT.control_side, _S.ZeroCrossAttn.control_side
_cache = {}


def _sft_side_nomlp(self, c):
    ch = _S.to_nhwc(c)
    key = ("actv", id(self))
    if key not in _cache:
        m = self.mlp_shared[0]
        _cache[key] = _S.ops.conv3x3(ch, m.w(), m.b32(), act=1)
    wgb, bgb = self._w_gamma_beta()
    return _S.ops.conv3x3(_cache[key], wgb, bgb)


def _side_cached(orig):
    def f(self, c):
        key = ("side", id(self))
        if key not in _cache:
            _cache[key] = orig(self, c)
        return _cache[key]
    return f


def configure(v):
    global retuned
    os.environ.pop("SUPIR_GN_APPLY", None)
    _S.ZeroSFT.control_side, _S.ZeroCrossAttn.control_side = _orig_sft_side, _orig_xattn_side
    if v == "nomlp":
        _S.ZeroSFT.control_side = _sft_side_nomlp
    if v == "noside":
        _S.ZeroSFT.control_side, _S.ZeroCrossAttn.control_side = _side_cached(_orig_sft_side), _side_cached(_orig_xattn_side)
    ops._TUNE.clear()
    ops._CHOICE.clear()
    if v == "retune":
        if retuned is not None:
            ops._TUNE.update(retuned[0])
            ops._CHOICE.update(retuned[1])
    else:
        ops._TUNE.update(file_tune)
        ops._CHOICE.update(file_choice)
    if v == "gnv1":
        os.environ["SUPIR_GN_APPLY"] = "v1"


res = {}
with torch.no_grad():
    for rep in range(2):
        for v in variants:
            configure(v)
            wrap.enable_graph(False)
            wrap._warm = False
            for _ in range(2):
                o = wrap(x, t, cond, 1.0)
            if v == "retune" and retuned is None:
                retuned = (dict(ops._TUNE), dict(ops._CHOICE))
                diff = {str(k): (file_tune.get(k), tl) for k, tl in ops._TUNE.items() if file_tune.get(k) != tl}
                print("retune picks that differ from the file (file, here):", json.dumps(diff))
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
            print(f"rep{rep} {v}: {ms:.2f} ms/step", flush=True)
    wrap.enable_graph(False)
os.environ.pop("SUPIR_GN_APPLY", None)
print(json.dumps(res))
