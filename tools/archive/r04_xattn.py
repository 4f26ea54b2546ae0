"""Isolated timing of the fused to_q + text cross-attention launch (csrc/xattn.hip) against the two launches it replaces, at the two
shapes of a CFG-doubled 1024^2 step; back to back (hot) and with a 64 MB cache-flushing copy between repetitions (cold weights)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from supir_amd.weights import fold_layernorm

dev, BF = "cuda", torch.bfloat16
res = {}
for (B, H, T, Tk, C) in [(2, 20, 1024, 77, 1280), (2, 10, 4096, 77, 640)]:
    N = H * 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, C, generator=g).to(dev).to(BF)
    k = torch.randn(B, Tk, N, generator=g).to(dev).to(BF)
    vt = torch.zeros(B, N, 128, dtype=BF, device=dev)
    vt[:, :, :Tk] = torch.randn(B, N, Tk, generator=g).to(dev).to(BF)
    w = torch.randn(N, C, generator=g).to(dev) * C ** -0.5
    wp = (torch.randn(C, C, generator=g).to(dev) * C ** -0.5).to(BF)
    xs, st = ops.gemm_ln(x.view(B * T, C), wp, None, residual=x.view(B * T, C), emit_stats=True)
    xs = xs.view(B, T, C)
    wf, cs, bf_ = fold_layernorm(w, None, torch.ones(C, device=dev), torch.zeros(C, device=dev) + 0.1)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def two():
        return ops.flash_attn(ops.gemm_ln(xs, wf, bf_, ln=st, colsum=cs), k, vt, B, H, T, Tk)

    def one():
        return ops.xattn_q(xs, wf, bf_, k, vt, B, H, T, Tk, ln=st, colsum=cs)

    for name, fn in (("two_launches", two), ("fused", one)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        hot = e0.elapsed_time(e1) / 50 * 1e3
        cold = []
        for _ in range(10):
            flush.add_(1)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        res[f"{(B, H, T, Tk, C)} {name}"] = dict(hot_us=round(hot, 2), cold_us=round(sorted(cold)[len(cold) // 2], 2))
        print(f"{(B, H, T, Tk, C)} {name}: hot {hot:.2f} us, cold (median of 10) {sorted(cold)[5]:.2f} us", flush=True)
    err = (one().float() - two().float()).norm() / two().float().norm()
    print(f"{(B, H, T, Tk, C)} fused vs two launches rel-L2 {err.item():.3e}", flush=True)
print(json.dumps(res))
