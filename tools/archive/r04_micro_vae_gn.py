"""Round-4 micro-benchmark of the VAE's GroupNorms (Normalize, sgm/modules/diffusionmodules/model.py:48-51) at the feature-map sizes of a
1024^2 image: statistics pass + apply pass (two launches) vs the producer-statistics form (supir_groupnorm_parts_finalize + apply with
`given`), and the whole VAE tail (denoise-encode + decode + encode + decode) with the producer statistics on / off.
Usage: python tools/r04_micro_vae_gn.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

BF, dev = torch.bfloat16, "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / iters * 1e3
        best = dt if best is None or dt < best else best
    return best


for (H, C) in [(1024, 128), (1024, 256), (512, 128), (512, 256), (512, 512), (256, 256), (256, 512), (128, 512)]:
    x = torch.randn(1, H, H, C, device=dev).to(BF)
    g, b = torch.randn(C, device=dev) * 0.2 + 1.0, torch.randn(C, device=dev) * 0.2
    out = torch.empty_like(x)
    nchunk = H * H // 256
    part = ops.GnPart(torch.rand(1, nchunk, C // 4, 2, device=dev), nchunk, C, 4)
    given = torch.rand(1, 32, 2, device=dev)
    row = {"HW": H * H, "C": C, "MB": round(x.numel() * 2 / 1e6, 1)}
    row["two_launch_us"] = round(timeit(lambda: ops.groupnorm(x, g, b, 1e-6, silu=True, out=out)), 1)
    row["stats_only_us"] = round(timeit(lambda: ops.groupnorm_stats(x)), 1)
    row["apply_given_us"] = round(timeit(lambda: ops.groupnorm(x, g, b, 1e-6, silu=True, out=out, given=given)), 1)
    row["finalize_plus_apply_us"] = round(timeit(lambda: ops.groupnorm(x, g, b, 1e-6, silu=True, out=out, part=part)), 1)
    row["apply_TBps"] = round(2 * x.numel() * 2 / row["apply_given_us"] / 1e6, 2)
    row["stats_TBps"] = round(x.numel() * 2 / row["stats_only_us"] / 1e6, 2)
    print(json.dumps(row), flush=True)
    del x, out
    torch.cuda.empty_cache()

# the whole tail, producer statistics on / off, interleaved
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.helpers import build_vae   # noqa: E402
from supir_amd.synth import synth_tensor   # noqa: E402

# the packaged picks predate tiles 39 / 40: let the VAE's shapes be timed again on this box
for k in [k for k in ops._TUNE if k[0] in ("conv", "gemm") and (k[5] if k[0] == "conv" else k[2]) in (128, 256, 512)]:
    del ops._TUNE[k]
vae = build_vae(dev)
img = synth_tensor("bench.vae", (1, 3, 1024, 1024), scale=0.5).clamp(-1, 1).to(dev)


def tail():
    with torch.no_grad():
        z = vae.quant_conv(vae.denoise_encoder(img))[:, :4]
        x1 = vae.decoder(vae.post_quant_conv(z))
        z1 = vae.quant_conv(vae.encoder(x1))[:, :4]
        return vae.decoder(vae.post_quant_conv(z1))


res = {}
for rep in range(3):
    for flag in (False, True):
        ops.USE_GN_PARTS = flag
        tail()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            o = tail()
        torch.cuda.synchronize()
        res.setdefault("parts_on" if flag else "parts_off", []).append(round((time.perf_counter() - t0) / 3 * 1e3, 2))
ops.USE_GN_PARTS = True
a = tail().float()
ops.USE_GN_PARTS = False
b_ = tail().float()
res["rel_l2_on_vs_off"] = ((a - b_).norm() / b_.norm()).item()
tr = ops.start_trace()
ops.USE_GN_PARTS = True
tail()
ops.stop_trace()
res["groupnorms"] = sum(1 for r in tr if r["kernel"] == "groupnorm")
res["from_producer"] = sum(1 for r in tr if r["kernel"] == "groupnorm_parts_finalize")
import collections
res["conv_tiles"] = dict(collections.Counter(r.get("tile") for r in tr if r["kernel"] == "conv3x3"))
print(json.dumps({"vae_tail_ms_2enc_2dec_1024px": res}))
