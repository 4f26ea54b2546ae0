import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from supir_amd import ops
BF = torch.bfloat16
for (B, Cin, H, W, Cout) in [(1, 128, 1024, 1024, 3), (1, 128, 512, 512, 3)]:
    x = torch.randn(B, H, W, Cin, device="cuda").to(BF)
    w9 = (torch.randn(9, Cout, Cin, device="cuda") * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda")
    for _ in range(3):
        ops.conv3x3_smallcout(x, w9, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3_smallcout(x, w9, b)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(json.dumps({"shape": [B, Cin, H, W, Cout], "us": round(us, 1), "input_TBps": round(B * H * W * Cin * 2 / us / 1e6, 2)}), flush=True)
