"""Tiny driver for rocprofv3 --pmc passes: a handful of launches of the hot kernels at production shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from supir_amd import ops
BF = torch.bfloat16
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
torch.manual_seed(0)
if which == "gemm":
    for (M, N, K, tile) in [(2048, 1280, 1280, 3), (2048, 1280, 1280, 1), (2048, 10240, 1280, 0), (2048, 1280, 5120, 3),
                            (8192, 640, 2560, 0)]:
        a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        for _ in range(3):
            ops.gemm(a, w, None, tile=tile)
elif which == "conv":
    for (B, H, W, Cin, Cout, tile) in [(2, 32, 32, 1280, 1280, 1), (2, 64, 64, 640, 640, 0), (2, 128, 128, 320, 320, 0)]:
        x = torch.randn(B, H, W, Cin, device=dev).to(BF); w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
        for _ in range(3):
            ops.conv3x3(x, w, None, tile=tile)
elif which == "attn":
    for (B, H, Tq, Tk) in [(2, 20, 1024, 1024), (2, 10, 4096, 4096)]:
        C = H * 64
        q = torch.randn(B, Tq, C, device=dev).to(BF); k = torch.randn(B, Tk, C, device=dev).to(BF)
        vt = torch.randn(B, C, Tk, device=dev).to(BF)
        for _ in range(3):
            ops.flash_attn(q, k, vt, B, H, Tq, Tk)
torch.cuda.synchronize()
