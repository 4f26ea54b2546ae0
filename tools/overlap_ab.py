"""Replayed 1024^2 step with the two free-running chains (GLVControl || UNet encoder on two HIP streams) vs ONE stream, interleaved, in
one process, shipped picks -- re-measured late in round 6 because the step turned out to be power-limited (profiles/r06/power_during_bench.json):
overlap cannot add power headroom, only fill launch tails.   python tools/overlap_ab.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests.helpers import build_unet, synth_tensor

dev = "cuda"
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
res, outs = {}, {}
with torch.no_grad():
    for rep in range(3):
        for name, ov in (("two_streams", True), ("one_stream", False)):
            wrap.enable_graph(False)
            wrap.overlap_branches = ov
            for _ in range(2):
                o = wrap(x, t, cond, 1.0)
            wrap.enable_graph(True)
            for _ in range(3):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            n = 20
            t0 = time.time()
            for _ in range(n):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            res.setdefault(name, []).append(round(ms, 3))
            outs[name] = o.float().clone()
            print(f"rep{rep} {name}: {ms:.3f} ms/step", flush=True)
    wrap.enable_graph(False)
    wrap.overlap_branches = True
eq = bool(torch.equal(outs["two_streams"], outs["one_stream"]))
print("outputs bitwise equal:", eq)
out = {"what": "replayed 1024^2 CFG-doubled step (ms), shipped picks, interleaved", "ms_per_step": res, "bitwise_equal": eq}
print(json.dumps(out))
go = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(go):
    json.dump(out, open(os.path.join(go, "overlap_ab.json"), "w"), indent=1)
