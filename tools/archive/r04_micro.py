"""Round-4 micro-benchmarks (hot, HIP events on the launch stream, one box, interleaved):
  * the 128 x 80 tile with eight waves / two K groups (35) vs four waves / one K group / 78 KB (38), plain GEMMs and 3x3 convolutions;
  * the 256 x 320 GEGLU tile with the exact-erf epilogue (debug knob 0 = 1) vs the fitted GELU (0), plus the error of both vs fp32 torch.
Usage: python tools/r04_micro.py"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib, ops
from supir_amd.weights import interleave_geglu

BF, dev = torch.bfloat16, "cuda"
lib = _lib.load()
lib.supir_debug_knob.argtypes = [ctypes.c_int, ctypes.c_int]
lib.supir_debug_knob.restype = ctypes.c_int


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / iters * 1e3
        best = dt if best is None or dt < best else best
    return best


out = {"gemm": [], "conv": [], "geglu": []}
for (M, N, K) in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 1280, 640), (8192, 640, 640), (2048, 1280, 2560)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(BF)
    row = {"M": M, "N": N, "K": K}
    for rep in range(2):
        for tile in (35, 38):
            row.setdefault(f"tile{tile}_us", []).append(round(timeit(lambda: ops.gemm(a, w, b, residual=res, tile=tile)), 2))
    ref = a.float() @ w.float().T + b + res.float()
    row["tile38_rel_l2"] = ((ops.gemm(a, w, b, residual=res, tile=38).float() - ref).norm() / ref.norm()).item()
    row["tflops_35_38"] = [round(2.0 * M * N * K / min(row[f"tile{t}_us"]) / 1e6, 1) for t in (35, 38)]
    out["gemm"].append(row)
    print(row, flush=True)
for (B, H, W, Cin, Cout) in [(2, 32, 32, 1280, 1280), (2, 32, 32, 2560, 1280), (2, 64, 64, 640, 640)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev)
    row = {"B": B, "H": H, "W": W, "Cin": Cin, "Cout": Cout}
    for rep in range(2):
        for tile in (35, 38):
            try:
                row.setdefault(f"tile{tile}_us", []).append(round(timeit(lambda: ops.conv3x3(x, w, b, tile=tile), iters=12), 2))
            except Exception as e:
                row[f"tile{tile}_us"] = repr(e)[:80]
    out["conv"].append(row)
    print(row, flush=True)
for (M, N2, K) in [(2048, 10240, 1280), (8192, 5120, 640)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N2, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N2, device=dev) * 0.1
    w16, b16 = interleave_geglu(w, b, 16)
    y = a.float() @ w.float().T + b
    n = N2 // 2
    ref = y[:, :n] * torch.nn.functional.gelu(y[:, n:])
    row = {"M": M, "N": N2, "K": K}
    for rep in range(2):
        for knob, name in ((1, "erf"), (0, "fit")):
            lib.supir_debug_knob(0, knob)
            row.setdefault(f"{name}_us", []).append(round(timeit(lambda: ops.gemm(a, w16, b16, act=2, tile=37), iters=20), 2))
            got = ops.gemm(a, w16, b16, act=2, tile=37).float()
            row[f"{name}_rel_l2_vs_fp32"] = ((got - ref).norm() / ref.norm()).item()
            row[f"{name}_max_abs_err"] = (got - ref).abs().max().item()
    lib.supir_debug_knob(0, 0)
    out["geglu"].append(row)
    print(row, flush=True)
print(json.dumps(out))
