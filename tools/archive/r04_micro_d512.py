"""Head-dim-512 attention (VAE mid block): key-split sweep -- one pass per workgroup vs 2..16 key splits vs the library's choice vs the
materialised-score form, at the token counts the five configs produce: 16 384 (1024^2 image), 4096 (512^2 image), 5184 / 7396 (tiled-VAE
encoder / decoder tiles of config 3, stacked per shape group), 1024 (256^2, the test suite).  Prints one JSON line per shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

dev, BF = "cuda", torch.bfloat16


def timed(fn, n=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


for B, T in ((1, 16384), (1, 4096), (4, 4096), (16, 4096), (1, 5184), (8, 5184), (1, 7396), (8, 7396), (1, 1024), (16, 1024)):
    g = torch.Generator(device="cpu").manual_seed(B * 100003 + T)
    q, k, v = (torch.randn(B, T, 512, generator=g).to(dev).to(BF) for _ in range(3))
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B, 512, Tp, dtype=BF, device=dev)
    vt[:, :, :T] = v.permute(0, 2, 1)
    row = {"B": B, "T": T}
    for s in (1, 2, 4, 8, 16, 0):
        row["auto" if s == 0 else f"x{s}"] = timed(lambda: ops.flash_attn_d512(q, k, vt, T, splits=s))

    def materialised():
        for b in range(B):
            sc = ops.gemm(q[b], k[b], out_dtype=torch.float32)
            p = ops.softmax_rows(sc, 512 ** -0.5, valid=T)
            ops.gemm(p, vt[b])

    if T % 64 == 0 and B * T * T <= 16384 * 16384:   # (the module pads ragged key counts to 64 for this form; not timed here)
        row["materialised"] = timed(materialised, n=3)
    row["tflops_auto"] = round(4.0 * B * T * T * 512 / row["auto"] / 1e6)
    print(json.dumps(row), flush=True)
