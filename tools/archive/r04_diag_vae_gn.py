"""Diagnostic (round 4): for every GroupNorm of a VAE pass whose input carries producer partials, compare the group statistics three ways
-- fp64 torch over the tensor (truth), the two-launch path's own statistics pass (ops.groupnorm_stats), the producer partials reduced by
supir_groupnorm_parts_finalize -- and report the relative error of rstd (what the normalised output scales with) and |mean| / std.
Also: the end-to-end difference between the two product paths next to the difference each has from a run whose statistics are the fp64
truth.  Usage: python tools/r04_diag_vae_gn.py [px]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib, ops
from tests.helpers import build_vae
from supir_amd.synth import synth_tensor

dev = "cuda"
px = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for k in [k for k in ops._TUNE if k[0] in ("conv", "gemm") and (k[5] if k[0] == "conv" else k[2]) in (128, 256, 512)]:
    del ops._TUNE[k]
vae = build_vae(dev)
img = synth_tensor("bench.vae", (1, 3, px, px), scale=0.5).clamp(-1, 1).to(dev)
real_gn = ops.groupnorm
rows = []
MODE = "diag"


def truth(x, eps):
    B, C = x.shape[0], x.shape[-1]
    g = x.double().reshape(B, -1, 32, C // 32)
    mean = g.mean(dim=(1, 3))
    var = (g * g).mean(dim=(1, 3)) - mean * mean
    return mean, var.clamp_min(0)


def gn(x, gamma, beta, eps, **kw):
    part = kw.get("part")
    if MODE == "truth":
        kw = dict(kw)
        kw.pop("part", None)
        m, v = truth(x, eps)
        return real_gn(x, gamma, beta, eps, given=torch.stack([m, v], -1).float().contiguous(), **kw)
    if MODE == "diag" and part is not None and part.unit == 4:
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        m, v = truth(x, eps)
        s = ops.groupnorm_stats(x).double()
        m2 = s[..., 0] / (HW * (C // 32))
        v2 = (s[..., 1] / (HW * (C // 32)) - m2 * m2).clamp_min(0)
        giv = torch.empty(B, 32, 2, dtype=torch.float32, device=x.device)
        lib = _lib.load(x.dtype)
        _lib.check(lib.supir_groupnorm_parts_finalize(part.buf.data_ptr(), B, part.nchunk, C, part.unit, HW, giv.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "fin", lib)
        m3, v3 = giv[..., 0].double(), giv[..., 1].double()
        r = lambda vv: 1.0 / torch.sqrt(vv + eps)
        rows.append({"HW": HW, "C": C, "mean_over_std_max": (m.abs() / v.sqrt()).max().item(),
                     "rstd_err_two_pass": ((r(v2) - r(v)).abs() / r(v)).max().item(),
                     "rstd_err_partials": ((r(v3) - r(v)).abs() / r(v)).max().item(),
                     "mean_err_two_pass": ((m2 - m).abs() / v.sqrt()).max().item(),
                     "mean_err_partials": ((m3 - m).abs() / v.sqrt()).max().item()})
    return real_gn(x, gamma, beta, eps, **kw)


ops.groupnorm = gn


def run():
    with torch.no_grad():
        z = vae.quant_conv(vae.denoise_encoder(img))[:, :4]
        return vae.decoder(vae.post_quant_conv(z)).float()


run()
rows.clear()
MODE = "diag"
out_parts = run()
for r in rows:
    print(json.dumps({k: (round(v, 9) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
MODE = "plain"
ops.USE_GN_PARTS = True
a = run()
ops.USE_GN_PARTS = False
b = run()
MODE = "truth"
t = run()
rl = lambda p, q: ((p - q).norm() / q.norm()).item()
print(json.dumps({"px": px, "partials_vs_two_pass": rl(a, b), "partials_vs_fp64_statistics": rl(a, t), "two_pass_vs_fp64_statistics": rl(b, t),
                  "two_pass_repeat": rl(b, (lambda: (setattr(ops, "USE_GN_PARTS", False), run())[1])())}))
