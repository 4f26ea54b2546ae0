"""Round-4 micro-benchmarks, second set (hot, one box, interleaved): the 256 x 160 tile with its eight waves as 8 x 1 (32 x 160 per
wave, knob 1 = 0) vs 4 x 2 (64 x 80 per wave, knob 1 = 1) -- same tile, same K order, so outputs must be BITWISE equal --, and the
GroupNorm apply launch with its default chunking vs one / two row batches per workgroup (knob 2).  Usage: python tools/r04_micro2.py"""
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


def timeit(fn, iters=20, warm=4):
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


def ab(name, fn, flops, knob, values=(0, 1), outs=None):
    row = {"case": name}
    got = {}
    for rep in range(2):
        for v in values:
            lib.supir_debug_knob(knob, v)
            row.setdefault(f"k{v}_us", []).append(round(timeit(fn), 2))
            o = fn()
            got[v] = [t.clone() for t in (o if isinstance(o, tuple) else (o,)) if torch.is_tensor(t)]
    lib.supir_debug_knob(knob, 0)
    row["bitwise_equal"] = all(all(torch.equal(a, b) for a, b in zip(got[values[0]], got[v])) for v in values[1:])
    if flops:
        row["tflops"] = [round(flops / min(row[f"k{v}_us"]) / 1e6, 1) for v in values]
    print(row, flush=True)
    return row


res = []
for (M, N, K) in [(8192, 640, 2560), (8192, 1280, 1280), (8192, 1280, 5120), (32768, 320, 320), (16384, 1280, 1280)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(BF)
    res.append(ab(f"gemm {M}x{N}x{K} tile34", lambda: ops.gemm(a, w, b, residual=r, tile=34), 2.0 * M * N * K, 1))
for (B, T, C) in [(2, 1024, 1280), (2, 4096, 640), (8, 1024, 1280)]:
    x = torch.randn(B, T, C, device=dev).to(BF)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(BF)
    b = torch.randn(3 * C, device=dev)
    if ops.gemm_qkv_supported(B * T, 3 * C, 2 * C, C, T):
        res.append(ab(f"qkv {B * T}x{3 * C}x{C}", lambda: ops.gemm_qkv(x, w, b, B, T, 2 * C), 2.0 * B * T * 3 * C * C, 1))
for (B, H, W, Cin, Cout) in [(2, 128, 128, 320, 320), (2, 64, 64, 1280, 1280), (1, 512, 512, 256, 256), (1, 1024, 1024, 128, 128),
                             (8, 32, 32, 1280, 1280)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev)
    try:
        res.append(ab(f"conv {B}x{H}x{W} {Cin}->{Cout} tile34", lambda: ops.conv3x3(x, w, b, tile=34), 2.0 * B * H * W * Cout * 9 * Cin, 1))
    except Exception as e:
        print("skip", (B, H, W, Cin, Cout), repr(e)[:100])
# GroupNorm apply chunking (statistics from a producer: one launch) at the UNet's shapes
for (B, HW, C) in [(2, 1024, 1280), (2, 4096, 640), (2, 16384, 320), (2, 1024, 2560)]:
    side = int(HW ** 0.5)
    a = torch.randn(B * HW, 320, device=dev).to(BF)
    w = (torch.randn(C, 320, device=dev) * 320 ** -0.5).to(BF)
    tile = 35 if C % 80 == 0 and (B * HW) % 128 == 0 else 33
    y, part = ops.gemm(a, w, None, rows_per_batch=HW, tile=tile, gn_part=True)
    y = y.view(B, side, side, C)
    g, be = torch.randn(C, device=dev) * 0.2 + 1.0, torch.randn(C, device=dev) * 0.1
    res.append(ab(f"groupnorm+silu {B}x{HW}x{C} parts={part is not None}", lambda: ops.groupnorm(y, g, be, 1e-5, silu=True, part=part), 0, 2,
                  values=(0, 1, 2)))
print(json.dumps(res))
