"""Round-4 micro-benchmark of the head-dim-64 flash attention: round-3 kernel (tools-only knob 3 = 1) vs the round-4 form (no tile
loads past the end, output staged through LDS for row-contiguous 16-byte stores).  Same arithmetic: outputs must be BITWISE equal.
Usage: python tools/r04_micro3.py"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib, ops

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


res = []
for (B, H, Tq, Tk) in [(2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (2, 10, 4096, 1024), (2, 20, 1000, 333)]:
    C = H * 64
    q = torch.randn(B, Tq, C, device=dev).to(BF)
    k = torch.randn(B, Tk, C, device=dev).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, device=dev, dtype=BF)
    vt[:, :, :Tk] = torch.randn(B, C, Tk, device=dev).to(BF)
    row = {"B": B, "H": H, "Tq": Tq, "Tk": Tk}
    outs = {}
    for rep in range(2):
        for knob, name in ((1, "r03"), (3, "r04"), (2, "r04_8waves")):
            lib.supir_debug_knob(3, knob)
            row.setdefault(f"{name}_us", []).append(round(timeit(lambda: ops.flash_attn(q, k, vt, B, H, Tq, Tk)), 2))
            outs[name] = ops.flash_attn(q, k, vt, B, H, Tq, Tk).clone()
    lib.supir_debug_knob(3, 0)
    row["bitwise_equal"] = bool(torch.equal(outs["r03"], outs["r04"]))
    qh = q.float().view(B, Tq, H, 64).permute(0, 2, 1, 3)
    kh = k.float().view(B, Tk, H, 64).permute(0, 2, 1, 3)
    vh = vt[:, :, :Tk].float().view(B, H, 64, Tk).transpose(-1, -2)
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, Tq, C)
    row["rel_l2_vs_fp32"] = ((outs["r04"].float() - ref).norm() / ref.norm()).item()
    fl = 4.0 * B * H * Tq * Tk * 64
    row["eight_waves_equal"] = bool(torch.equal(outs["r04"], outs["r04_8waves"]))
    row["tflops"] = [round(fl / min(row[f"{n}_us"]) / 1e6, 1) for n in ("r03", "r04", "r04_8waves")]
    res.append(row)
    print(row, flush=True)
print(json.dumps(res))
