"""Workload for rocprofv3 --kernel-trace --stats: N eager CFG-doubled UNet+control steps at 1024^2 (serial streams, so
per-kernel durations are not blurred by the two-stream overlap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import build_unet, synth_tensor
dev = "cuda"
lat = 128
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wrap = build_unet(device=dev)
wrap.overlap_branches = "--overlap" in sys.argv
B = 2
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.tensor([500, 500], dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(steps):
        wrap(x, t, cond, 1.0)
torch.cuda.synchronize()
from supir_amd import ops
ops.save_tuning()
