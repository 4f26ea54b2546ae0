"""Flash attention at the step's shapes: time per launch and error against an fp32 reference of the same op."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

BF = torch.bfloat16
torch.manual_seed(0)
for (B, H, Tq, Tk) in [(2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (2, 10, 4096, 1024),
                       (1, 1, 16384, 16384)]:
    C = H * 64
    q = torch.randn(B, Tq, C, device="cuda").to(BF)
    k = torch.randn(B, Tk, C, device="cuda").to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, device="cuda", dtype=BF)
    vt[:, :, :Tk] = torch.randn(B, C, Tk, device="cuda").to(BF)
    for _ in range(3):
        out = ops.flash_attn(q, k, vt, B, H, Tq, Tk)
    err = None
    if Tq * Tk <= 4096 * 4096 and B * H <= 40:   # fp32 reference of the same op
        qh = q.float().view(B, Tq, H, 64).transpose(1, 2)
        kh = k.float().view(B, Tk, H, 64).transpose(1, 2)
        vh = vt[:, :, :Tk].float().view(B, H, 64, Tk).transpose(2, 3)
        ref = torch.softmax(qh @ kh.transpose(2, 3) * 0.125, -1) @ vh
        d = out.float().view(B, Tq, H, 64).transpose(1, 2) - ref
        err = (round(float(d.abs().max() / ref.abs().max()), 5), round(float(d.norm() / ref.norm()), 5))   # (max / max, rel L2)
        del qh, kh, vh, ref
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.flash_attn(q, k, vt, B, H, Tq, Tk)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(dict(err=err, B=B, H=H, Tq=Tq, Tk=Tk, us=round(t * 1e6, 1),
               tflops=round(4.0 * B * H * Tq * Tk * 64 / t / 1e12, 1)), flush=True)
