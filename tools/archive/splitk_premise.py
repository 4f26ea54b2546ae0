import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from supir_amd import ops
BF = torch.bfloat16
def t_conv(B, H, Cin, Cout, tile):
    x = torch.randn(B, H, H, Cin, device="cuda").to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda")
    out = torch.empty(B, H, H, Cout, device="cuda", dtype=BF)
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            ops.conv3x3(x, w, b, out=out, tile=tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv3x3(x, w, b, out=out, tile=tile)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"conv B{B} {H}x{H} {Cin}->{Cout} tile {tile}: {best:.1f} us {fl / best / 1e6:.0f} TFLOP/s", flush=True)
def t_gemm(M, N, K, tile):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    res = torch.randn(M, N, device="cuda").to(BF)
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            ops.gemm(a, w, None, residual=res, tile=tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm(a, w, None, residual=res, tile=tile)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print(f"gemm M{M} N{N} K{K} tile {tile}: {best:.1f} us {2.0*M*N*K / best / 1e6:.0f} TFLOP/s", flush=True)
t_conv(2, 32, 1280, 1280, 35)      # today
t_conv(8, 32, 320, 1280, 34)       # 256 workgroups x 45 K steps = the main loops of a 4-way split of the line above on 256 x 160 tiles
t_conv(8, 32, 1280, 1280, 34)      # 256 workgroups x 180 K steps
t_conv(2, 32, 2560, 1280, 35)
t_conv(8, 32, 640, 1280, 34)
t_conv(2, 64, 640, 640, 33)        # today: 8192 x 640, K 5760
t_conv(4, 64, 320, 640, 34)        # 2-way split of it on 256 x 160: 256 workgroups x 45 K steps
t_conv(2, 64, 1280, 640, 33)
t_conv(4, 64, 640, 640, 34)
t_gemm(2048, 1280, 5120, 35)
t_gemm(8192, 1280, 1280, 34)       # 4-way split of the ff output projection: 256 workgroups x 20 K steps
t_gemm(2048, 1280, 1280, 35)
t_gemm(8192, 1280, 320, 34)
