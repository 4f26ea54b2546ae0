"""Workload for `rocprofv3 --kernel-trace`: capture the 1024^2 CFG-doubled step into a hipGraph, then replay it a few times with
idle gaps in between so that tools/analyze_trace.py can cut the trace into steps.  Usage (from /tmp, TMPDIR=/tmp):
  rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/trace_step.py [gemm16|base]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from tests.helpers import build_unet, synth_tensor

ops.USE_GEMM16 = (sys.argv[1] if len(sys.argv) > 1 else "gemm16") == "gemm16"
dev = "cuda"
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
        "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(2):
        wrap(x, t, cond, 1.0)
    wrap.enable_graph(True)
    for _ in range(3):
        wrap(x, t, cond, 1.0)
    torch.cuda.synchronize()
    for _ in range(3):
        time.sleep(0.2)
        t0 = time.time()
        wrap(x, t, cond, 1.0)
        torch.cuda.synchronize()
        print(f"replay {1e3 * (time.time() - t0):.2f} ms", flush=True)
    time.sleep(0.2)
    t0 = time.time()
    for _ in range(5):
        wrap(x, t, cond, 1.0)
    torch.cuda.synchronize()
    print(f"5 back-to-back replays {1e3 * (time.time() - t0) / 5:.2f} ms/step", flush=True)
