"""rocprofv3 --pmc driver (round 6): ONE convolution (1280 -> 1280 @ 32 x 32, B = 2) and one GEMM on the loop forms of the 128 x 80 tile --
35 (implicit GEMM, one barrier per step), 48 (LDS-staged halo), 57 (software-pipelined fragments), 58 (halo + software pipeline), 54 / 55
(ping-pong K groups) -- three launches each, in this order; the counter CSV is grouped by kernel name.  tools/pmc_forms_summarize.py reads it."""
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
for tile in (35, 48, 57, 58, 54, 55):
    for _ in range(3):
        ops.conv3x3(x, w, None, tile=tile)
x2 = torch.randn(2, 64, 64, 640, device=dev).to(BF)
w2 = (torch.randn(640, 3, 3, 640, device=dev) * (9 * 640) ** -0.5).to(BF)
for tile in (33, 49):
    for _ in range(3):
        ops.conv3x3(x2, w2, None, tile=tile)
torch.cuda.synchronize()
