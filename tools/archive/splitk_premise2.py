"""Premise check for a TWO-way split-K of the long-K M = 2048 launches on 128 x 160 tiles (tile 33): the main loops alone, timed as
full-machine launches of the same per-workgroup work (M doubled, K halved: 256 workgroups x K/2), next to today's 128 x 80 launch and the
4-way / 256 x 160 proxy of tools/splitk_premise.py.  No reduction in these numbers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.splitk_premise import t_conv, t_gemm  # noqa: E402  (runs that script's lines first)

print("--- two-way split proxies on 128 x 160", flush=True)
t_gemm(2048, 1280, 5120, 35)       # today
t_gemm(4096, 1280, 2560, 33)       # 2-way split on 128 x 160: 256 workgroups x 40 K steps (20 per K group)
t_gemm(8192, 1280, 1280, 34)       # 4-way on 256 x 160
t_conv(2, 32, 1280, 1280, 35)      # today
t_conv(4, 32, 640, 1280, 33)       # 2-way on 128 x 160
t_conv(2, 32, 2560, 1280, 35)      # today
t_conv(4, 32, 1280, 1280, 33)      # 2-way on 128 x 160
t_conv(2, 32, 1920, 1280, 35)
t_conv(2, 32, 640, 1280, 35)
