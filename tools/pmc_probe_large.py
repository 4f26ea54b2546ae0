"""rocprofv3 --pmc driver for the LARGE-M regime (round 5): the tiles that carry config 3 (tile batches: M = 8192 and
up), `--num_samples 4` and the VAE tail -- tiles 39 / 40 / 41 on the VAE's convolutions, 256 x 160 (tile 34) and 256 x 320 (tile 37) at
M >= 8192.  Same conventions as tools/pmc_probe.py: every case is launched three times, cases in a fixed order; tools/pmc_summarize.py maps
(kernel name, grid size, position) back to the case list this script writes."""
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
ONLY = set(int(t) for t in os.environ.get("PMC_TILES", "").split(",") if t)


def note(kind, shape, flops, bytes_, tile):
    cases.append(dict(kind=kind, shape=shape, flops=flops, algorithmic_bytes=bytes_, tile=tile))


def want(tile):
    return (not ONLY or tile in ONLY) and (tile in ops._G16 or tile == 37 or tile < 8 or tile in getattr(ops, "_G256", {}))


# VAE convolutions (sgm/modules/diffusionmodules/model.py:55-148 at 1024^2): Cin -> Cout at H x W
for (H, Cin, Cout, tiles) in [(256, 512, 512, (40, 39, 42)), (512, 256, 256, (40, 39, 42)), (512, 256, 128, (39,)), (1024, 128, 128, (39,))]:
    x = torch.randn(1, H, H, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    for tile in tiles:
        if not want(tile) or (tile in (40, 41) and Cout % 256):
            continue
        for _ in range(3):
            ops.conv3x3(x, w, None, tile=tile)
        note("conv3x3", f"B1 {H}x{H} {Cin}->{Cout}", 2.0 * H * H * Cout * 9 * Cin, 2.0 * (H * H * (Cin + Cout) + Cout * 9 * Cin), tile)
    del x, w
# M = 8192 (tile batch 4 / num_samples 4 at the 32 x 32 level) and M = 32768 (64 x 64 level) GEMMs
for (M, N, K, tiles) in [(8192, 1280, 1280, (34, 33)), (8192, 1280, 5120, (34, 33)), (32768, 640, 640, (34, 33)), (32768, 640, 2560, (34,))]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    for tile in tiles:
        if not want(tile):
            continue
        for _ in range(3):
            ops.gemm(a, w, None, residual=res, tile=tile)
        note("gemm", f"M{M} N{N} K{K}", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N), tile)
for (M, K, N2) in [(8192, 1280, 10240), (32768, 640, 5120)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N2, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N2, device=dev)
    w16, b16 = interleave_geglu(w, b, 16)
    for tile in (37, 34):
        if not want(tile):
            continue
        for _ in range(3):
            ops.gemm(a, w16, b16, act=2, tile=tile)
        note("gemm_geglu", f"M{M} N{N2} K{K}", 2.0 * M * N2 * K, 2.0 * (M * K + N2 * K + M * N2 // 2), tile)
# UNet convolutions at batch 8
for (B, H, Cin, Cout, tile) in [(8, 32, 1280, 1280, 34), (8, 64, 640, 640, 34), (8, 128, 320, 320, 34)]:
    if not want(tile):
        continue
    x = torch.randn(B, H, H, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    for _ in range(3):
        ops.conv3x3(x, w, None, tile=tile)
    note("conv3x3", f"B{B} {H}x{H} {Cin}->{Cout}", 2.0 * B * H * H * Cout * 9 * Cin, 2.0 * (B * H * H * (Cin + Cout) + Cout * 9 * Cin), tile)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    json.dump(cases, open(sys.argv[1], "w"), indent=1)
