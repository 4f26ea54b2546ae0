"""In-process A/B of the small-Cin boundary convolution (supir_conv3x3_smallcin): the exact-fp32 matrix-instruction form
(conv3x3_smallcin_mfma_kernel, round 5) against the VALU form it replaces (tools knob 7), interleaved on one box, on the path's shapes:
the UNet / control input convolutions (4 -> 320 at the latent resolution, CFG batch 2 and tile batch 8), the hint convolution with its
fused add, the VAE encoder's conv_in (3 -> 128 at image resolution) and the decoder's (4 -> 512 at latent resolution).

    python tools/bench_smallcin.py [out.json]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supir_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
CASES = [("unet_in 4->320 @128^2 B2", 2, 4, 128, 128, 320, False), ("hint 4->320 @128^2 B2 +add", 2, 4, 128, 128, 320, True),
         ("unet_in 4->320 @128^2 B8", 8, 4, 128, 128, 320, False), ("vae_enc 3->128 @1024^2", 1, 3, 1024, 1024, 128, False),
         ("vae_dec 4->512 @128^2", 1, 4, 128, 128, 512, False), ("vae_enc tile 3->128 @576^2", 1, 3, 576, 576, 128, False)]


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    lib = _lib.load(BF)
    g = torch.Generator(device="cuda").manual_seed(0)
    res = {}
    for name, B, Cin, H, W, Cout, with_add in CASES:
        x = torch.randn(B, Cin, H, W, device="cuda", generator=g)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (9 * Cin) ** -0.5
        b = torch.randn(Cout, device="cuda", generator=g)
        add = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(BF) if with_add else None
        out = torch.empty(B, H, W, Cout, dtype=BF, device="cuda")
        run = lambda: ops.conv3x3_smallcin(x, w, b, add=add, out=out)
        us = {"mfma": [], "valu": []}
        outs = {}
        for rep in range(3):
            for form, knob in (("mfma", 0), ("valu", 1)):
                with _lib.tools_knob(7, knob):      # both arms on libsupir_hip_tools.so (the only build with variant switches)
                    for _ in range(3):
                        run()
                    us[form].append(round(timed(run, 20), 2))
                    outs[form] = out.clone()
        d = (outs["mfma"].float() - outs["valu"].float())
        flop = 2.0 * B * H * W * Cout * 9 * Cin
        byt = B * H * W * (4.0 * Cin + 2.0 * Cout * (2 if with_add else 1))
        m = min(us["mfma"])
        res[name] = dict(us_mfma=us["mfma"], us_valu=us["valu"], speedup=round(min(us["valu"]) / m, 2), tflops_mfma=round(flop / m / 1e6, 1),
                         gbps_mfma=round(byt / m / 1e3, 1), frac_hbm_peak=round(byt / m / 1e3 / 8000.0, 3),
                         rel_l2_between_forms=float(d.norm() / outs["valu"].float().norm()), elements_differing=float((d != 0).float().mean()))
        print(name, res[name], flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
