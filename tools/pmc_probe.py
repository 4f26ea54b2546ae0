"""Tiny driver for rocprofv3 --pmc passes: a handful of launches of the hot kernels at production shapes (round 2: the
16x16x32-MFMA tile family of csrc/gemm16.hip, the implicit-GEMM convs on it, attention, GroupNorm).  Each case is tagged by its
grid size in the counter CSV; tools/pmc_summarize.py maps (kernel name, grid) back to the case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from supir_amd.weights import interleave_geglu

BF = torch.bfloat16
dev = "cuda"
torch.manual_seed(0)
cases = []


def note(kind, shape, flops, bytes_, tile):
    cases.append(dict(kind=kind, shape=shape, flops=flops, algorithmic_bytes=bytes_, tile=tile))


# GEGLU projection on tile 34 (the dominant kernel of the step), N = 1280 GEMMs on tile 35, q|k on tile 33
M, K, N2 = 2048, 1280, 10240
a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N2, K, device=dev) * K ** -0.5).to(BF); b = torch.randn(N2, device=dev)
w16, b16 = interleave_geglu(w, b, 16)
for _ in range(3):
    ops.gemm(a, w16, b16, act=2, tile=34)
note("gemm_geglu", f"M{M} N{N2} K{K}", 2.0 * M * N2 * K, 2.0 * (M * K + N2 * K + M * N2 // 2), 34)
for _ in range(3):   # the same projection on the 256 x 320 tile of csrc/gemm_big.hip (grid 256 instead of 512)
    ops.gemm(a, w16, b16, act=2, tile=37)
note("gemm_geglu", f"M{M} N{N2} K{K}", 2.0 * M * N2 * K, 2.0 * (M * K + N2 * K + M * N2 // 2), 37)
for (M, N, K, tile) in [(2048, 1280, 1280, 35), (2048, 1280, 5120, 35), (2048, 2560, 1280, 33), (8192, 640, 2560, 33)]:
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    for _ in range(3):
        ops.gemm(a, w, None, residual=res, tile=tile)
    note("gemm", f"M{M} N{N} K{K}", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N), tile)
# fused q|k|v projection of the 1280-wide blocks: 240 tiles of 256 x 128 (round 4)
M, N, K = 2048, 3840, 1280
a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
for _ in range(3):
    ops.gemm_qkv(a, w, None, 2, 1024, 2560)
note("gemm_qkv", f"M{M} N{N} K{K}", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), 41)
for (B, H, W, Cin, Cout, tile) in [(2, 32, 32, 1280, 1280, 35), (2, 64, 64, 640, 640, 33), (2, 128, 128, 320, 320, 34)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF); w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    for _ in range(3):
        ops.conv3x3(x, w, None, tile=tile)
    note("conv3x3", f"B{B} {H}x{W} {Cin}->{Cout}", 2.0 * B * H * W * Cout * 9 * Cin, 2.0 * (B * H * W * (Cin + Cout) + Cout * 9 * Cin), tile)
for (B, H, Tq, Tk) in [(2, 20, 1024, 1024), (2, 10, 4096, 4096)]:
    C = H * 64
    q = torch.randn(B, Tq, C, device=dev).to(BF); k = torch.randn(B, Tk, C, device=dev).to(BF)
    vt = torch.randn(B, C, Tk, device=dev).to(BF)
    for _ in range(3):
        ops.flash_attn(q, k, vt, B, H, Tq, Tk)
    note("attn", f"B{B} H{H} Tq{Tq} Tk{Tk}", 4.0 * B * H * Tq * Tk * 64, 2.0 * B * C * (2 * Tq + 2 * Tk), -1)
# fused to_q + text cross-attention (csrc/xattn.hip, round 4): x, the head's weight rows and the 77 cached keys in, the attention output out
for (B, H, T, Tk, C) in [(2, 20, 1024, 77, 1280), (2, 10, 4096, 77, 640)]:
    N = H * 64
    x = torch.randn(B, T, C, device=dev).to(BF); wq = (torch.randn(N, C, device=dev) * C ** -0.5).to(BF)
    k = torch.randn(B, Tk, N, device=dev).to(BF)
    vt = torch.zeros(B, N, 128, device=dev, dtype=BF); vt[:, :, :Tk] = torch.randn(B, N, Tk, device=dev).to(BF)
    for _ in range(3):
        ops.xattn_q(x, wq, None, k, vt, B, H, T, Tk)
    note("xattn_q", f"B{B} H{H} T{T} Tk{Tk} C{C}", 2.0 * B * T * N * C + 4.0 * B * T * N * Tk, 2.0 * (B * T * C + N * C + 2 * B * Tk * N + B * T * N), -1)
for (B, HW, C) in [(2, 1024, 1280), (2, 4096, 640), (2, 16384, 320)]:
    x = torch.randn(B, HW, C, device=dev).to(BF)
    g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(3):
        ops.groupnorm(x.view(B, HW, 1, C), g, bt, 1e-5, silu=True)
    note("groupnorm", f"B{B} HW{HW} C{C}", 0.0, 4.0 * B * HW * C, -1)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    json.dump(cases, open(sys.argv[1], "w"), indent=1)
