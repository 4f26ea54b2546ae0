"""Round-4 micro-benchmark of the VAE's 3x3 convolutions (sgm/modules/diffusionmodules/model.py:55-148, 571-743) at the sizes of a
1024^2 image: the gemm.hip tile the shipped tune file picks for each shape vs tiles 39 (256 x 128) / 40 (256 x 256) of csrc/gemm16.hip.
Prints one JSON row per shape (us per launch, TFLOP/s, relative L2 between the forms).  Usage: python tools/r04_micro_vae.py"""
import json
import os
import sys

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


# (B, H, W, Cin, Cout, stride, upsample): every "conv" key of the shipped tune file with VAE channel counts
keys = sorted(k for k in ops._TUNE if k[0] == "conv" and k[1] == 1 and k[4] in (128, 256, 512) and k[5] in (128, 256, 512) and len(k) == 8)
rows = []
for key in keys:
    _, B, H, W, Cin, Cout, stride, up = key
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    bias = torch.randn(Cout, device=dev)
    kw = dict(stride=stride, upsample=bool(up))
    if stride == 2:
        kw.update(pad=(0, 0), out_hw=(H // 2, W // 2))
    OH, OW = (2 * H, 2 * W) if up else ((H // 2, W // 2) if stride == 2 else (H, W))
    M = B * OH * OW
    fl = 2.0 * M * Cout * 9 * Cin
    shipped = ops._TUNE[key]
    row = {"shape": list(key[1:]), "M": M, "shipped_tile": shipped}
    outs = {}
    for t in [shipped, 4, 5, 39, 40]:
        if t in outs or (t == 39 and (M % 256 or Cout % 128)) or (t == 40 and (M % 256 or Cout % 256)):
            continue
        try:
            us = timeit(lambda: ops.conv3x3(x, w, bias, tile=t, **kw))
        except Exception as e:   # a forced tile the shape does not fit
            row[f"tile{t}"] = str(e)[:60]
            continue
        outs[t] = ops.conv3x3(x, w, bias, tile=t, **kw).float()
        row[f"tile{t}_us"] = round(us, 1)
        row[f"tile{t}_tflops"] = round(fl / us / 1e6, 1)
    base = outs[shipped]
    for t in (39, 40):
        if t in outs:
            row[f"tile{t}_rel_l2_vs_shipped"] = ((outs[t] - base).norm() / base.norm()).item()
    del outs, x, w
    torch.cuda.empty_cache()
    rows.append(row)
    print(json.dumps(row), flush=True)
best_old = sum(r[f"tile{r['shipped_tile']}_us"] for r in rows)
best_new = sum(min(v for k, v in r.items() if k.endswith("_us")) for r in rows)
print(json.dumps({"sum_us_shipped_picks": round(best_old, 1), "sum_us_best_of_all": round(best_new, 1)}))
