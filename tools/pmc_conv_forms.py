"""rocprofv3 --pmc driver (round 6): the UNet's stride-1 convolutions in the implicit-GEMM form and in the LDS-staged halo form of the same tile
(35 / 48 at 32 x 32, 33 / 49 at 64 x 64, 34 / 50 at B = 8) -- three launches each; the counter CSV is grouped by kernel name
(tools/pmc_forms_summarize.py).  (profiles/r06/pmc_loop_forms_summary.json is the run that also held the ping-pong / software-pipelined loops.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

BF, dev = torch.bfloat16, "cuda"
torch.manual_seed(0)
B, H, W, Cin, Cout = 2, 32, 32, 1280, 1280
x = torch.randn(B, H, W, Cin, device=dev).to(BF)
w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
for tile in (35, 48):
    for _ in range(3):
        ops.conv3x3(x, w, None, tile=tile)
x2 = torch.randn(2, 64, 64, 640, device=dev).to(BF)
w2 = (torch.randn(640, 3, 3, 640, device=dev) * (9 * 640) ** -0.5).to(BF)
for tile in (33, 49):
    for _ in range(3):
        ops.conv3x3(x2, w2, None, tile=tile)
x8 = torch.randn(8, H, W, Cin, device=dev).to(BF)
for tile in (34, 50):
    for _ in range(3):
        ops.conv3x3(x8, w, None, tile=tile)
torch.cuda.synchronize()
