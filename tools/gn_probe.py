"""GroupNorm (+SiLU) at the UNet's map sizes: microseconds per call (statistics + apply launches) and effective GB/s (read + write once).
SUPIR_LIB=<path> loads another build of the library (A/B on one box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib
if os.environ.get("SUPIR_LIB"):
    _lib.LIB_PATH = os.environ["SUPIR_LIB"]
from supir_amd import ops

if os.environ.get("SUPIR_KNOB2"):      # debug knob 2 of csrc/norm.hip (apply chunking), see csrc/kernels.h
    _ctx = _lib.tools_knob(2, int(os.environ["SUPIR_KNOB2"]))      # libsupir_hip_tools.so for the rest of the process
    _ctx.__enter__()
BF = torch.bfloat16
torch.manual_seed(0)
SHAPES = [(2, 16384, 320, 0), (2, 16384, 640, 320), (2, 4096, 640, 0), (2, 4096, 1280, 640), (2, 1024, 1280, 0), (2, 1024, 1280, 1280),
          (1, 1 << 20, 128, 0), (1, 1 << 18, 256, 0), (1, 1 << 16, 512, 0)]     # + the VAE's 1024^2 / 512^2 / 256^2 maps
for (B, HW, C, C2) in SHAPES:
    x = torch.randn(B, HW, C, device="cuda").to(BF)
    x2 = torch.randn(B, HW, C2, device="cuda").to(BF) if C2 else None
    g, b = torch.rand(C + C2, device="cuda") + 0.5, torch.randn(C + C2, device="cuda")
    out = torch.empty(B, HW, C + C2, device="cuda", dtype=BF)
    for _ in range(3):
        ops.groupnorm(x, g, b, 1e-5, silu=True, x2=x2, out=out)
    torch.cuda.synchronize()
    # 50 calls captured into one hipGraph: a Python loop of launches this small is CPU-bound (~9 us per launch) and measures nothing
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(50):
                ops.groupnorm(x, g, b, 1e-5, silu=True, x2=x2, out=out)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    xf = torch.cat([x, x2], -1).float() if C2 else x.float()
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(xf.transpose(1, 2), 32, g, b, 1e-5)).transpose(1, 2)
    err = float((out.float() - ref).norm() / ref.norm())
    print(dict(B=B, HW=HW, C=C + C2, concat=bool(C2), us=round(t * 1e6, 1), gbps=round(2 * 2.0 * B * HW * (C + C2) / t / 1e9), err=round(err, 5)), flush=True)
