"""Head-dim-64 flash attention: row sum as v_dot2c of the packed bf16 probabilities (default) vs the fp32 sum of the unrounded exponentials
(tools-only knob 3 = 4), same launch policy (eight waves where the launch is one round, four otherwise): microseconds, and the error of
both against fp32 SDPA on the same bf16 operands.  Usage: python tools/r04_micro_attn_ds.py"""
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


for (B, H, Tq, Tk, scale) in [(2, 20, 1024, 1024, 1.0), (2, 10, 4096, 4096, 1.0), (2, 10, 4096, 1024, 1.0), (2, 20, 1000, 333, 1.0),
                              (2, 20, 1024, 1024, 3.0), (2, 10, 4096, 4096, 3.0)]:
    C = H * 64
    g = torch.Generator(device="cpu").manual_seed(Tq * 7 + Tk)
    q = (torch.randn(B, Tq, C, generator=g) * scale).to(dev).to(BF)
    k = torch.randn(B, Tk, C, generator=g).to(dev).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, device=dev, dtype=BF)
    vt[:, :, :Tk] = torch.randn(B, C, Tk, generator=g).to(dev).to(BF)
    qh = q.float().view(B, Tq, H, 64).permute(0, 2, 1, 3)
    kh = k.float().view(B, Tk, H, 64).permute(0, 2, 1, 3)
    vh = vt[:, :, :Tk].float().view(B, H, 64, Tk).transpose(-1, -2)
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, Tq, C)
    row = {"B": B, "H": H, "Tq": Tq, "Tk": Tk, "q_scale": scale}
    for rep in range(2):
        for knob, name in ((4, "fp32_sum"), (0, "dot2_sum")):
            lib.supir_debug_knob(3, knob)
            row.setdefault(f"{name}_us", []).append(round(timeit(lambda: ops.flash_attn(q, k, vt, B, H, Tq, Tk)), 2))
            o = ops.flash_attn(q, k, vt, B, H, Tq, Tk).float()
            row[f"{name}_rel_l2_vs_fp32"] = ((o - ref).norm() / ref.norm()).item()
    lib.supir_debug_knob(3, 0)
    print(json.dumps(row), flush=True)
