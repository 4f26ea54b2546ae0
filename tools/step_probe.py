"""Time one CFG-doubled UNet+control step at 1024^2 (latent 128^2, B=2) eagerly and under hipGraph replay; dump the
per-kernel-class op trace (counts, algorithmic FLOPs / bytes)."""
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
lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
wrap = build_unet(device=dev)
torch.cuda.synchronize()
print("build+fill s", time.time() - t0, "mem GB", torch.cuda.memory_allocated() / 1e9, flush=True)
B = BATCH
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(2):
        out = wrap(x, t, cond, 1.0)
    torch.cuda.synchronize()
    print("out std", out.std().item(), "finite", torch.isfinite(out).all().item(), "mem GB", torch.cuda.memory_allocated() / 1e9)
    tr = ops.start_trace()
    wrap(x, t, cond, 1.0)
    tr = ops.stop_trace()
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in tr:
        a = agg[r["kernel"]]
        a[0] += 1; a[1] += r["flops"]; a[2] += r["bytes"]
    for k, (n, f, b) in agg.items():
        print(f"  {k:16s} launches {n:5d}  TFLOP {f / 1e12:8.3f}  GB {b / 1e9:8.3f}")
    print("  total launches", len(tr), "TFLOP", sum(r["flops"] for r in tr) / 1e12)
    for rep in range(2):
        for dist in (0, 1, 2, 3):
            wrap.prefetch_distance = dist
            wrap.overlap_branches = True
            wrap.enable_graph(False)
            wrap.enable_graph(True)
            for _ in range(3):
                wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            n = 8
            t1 = time.time()
            for _ in range(n):
                wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            print(f"rep{rep} prefetch distance {dist}: {(time.time() - t1) / n * 1e3:.2f} ms/step", flush=True)
    wrap.enable_graph(False)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(tr, open("gpurun_out/step_trace.json", "w"))
