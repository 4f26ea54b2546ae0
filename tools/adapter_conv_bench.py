"""The adapters' control-side convolutions (ZeroSFT.control_side: mlp_shared Cc -> 128 + SiLU, then 128 -> 2 * Ccat for gamma | beta) at the
shapes of a 1024^2 step: microseconds per candidate tile (hot, back to back), incl. the tap-split forms (64 + t).
Usage: python tools/adapter_conv_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

dev, BF = "cuda", torch.bfloat16
SHAPES = [  # (B, H, W, Cin, Cout, act)
    (2, 32, 32, 1280, 128, 1), (2, 32, 32, 640, 128, 1), (2, 64, 64, 640, 128, 1), (2, 64, 64, 320, 128, 1), (2, 128, 128, 320, 128, 1),
    (2, 32, 32, 128, 5120, 0), (2, 32, 32, 128, 2560, 0), (2, 64, 64, 128, 3840, 0), (2, 64, 64, 128, 2560, 0), (2, 128, 128, 128, 1920, 0),
    (2, 128, 128, 128, 1280, 0), (2, 128, 128, 128, 640, 0)]
for (B, H, W, Cin, Cout, act) in SHAPES:
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (9 * Cin) ** -0.5).to(dev).to(BF)
    bias = torch.randn(Cout, generator=g).to(dev)
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    res = []
    for t in [0, 1, 2, 3, 4, 5, 6, 32, 33, 34, 35, 65, 66, 67]:
        try:
            for _ in range(3):
                ops.conv3x3(x, w, bias, act=act, tile=t)
        except Exception:  # noqa: BLE001  (tile does not fit the shape)
            continue
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv3x3(x, w, bias, act=act, tile=t)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 100
            best = us if best is None or us < best else best
        res.append((best, t))
    res.sort()
    print(f"B{B} {H}x{W} {Cin}->{Cout} act{act}: " + "  ".join(f"t{t}:{us:.1f}us" for us, t in res[:8]) + f"   best {fl / res[0][0] / 1e6:.0f} TF/s",
          flush=True)
