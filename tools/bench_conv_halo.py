"""Within-process interleaved A/B of the stride-1 3x3 convolutions of the UNet / control net (round 6): the implicit-GEMM tiles that ran them
(32-35: every tap staged from L2) against the LDS-staged halo tiles 48-51 (csrc/gemm16.hip), at B = 2 (one image, CFG-doubled) and B = 8
(tile batch / num_samples 4).  Weights rotate over 8 copies (cold-ish: a step never finds a layer's weights in L2).  Rounds are interleaved
(tile A, tile B, ... repeated) so clock drift hits every variant alike; reports median / min microseconds and TFLOP/s, and max |a - b| between
forms.  Usage: python tools/bench_conv_halo.py out.json [--rounds 7]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

BF, dev = torch.bfloat16, "cuda"
out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/conv_halo_ab.json"
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 7


def ab(fns, iters):
    for f in fns.values():
        f()
        f()
    ts = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            e1.synchronize()
            ts[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in ts.items()}


HALO = {48: (128, 80, 2, 32), 49: (128, 160, 2, 64), 50: (256, 160, 1, 32), 51: (256, 160, 1, 64)}
OLD = {32: (128, 80, 2), 33: (128, 160, 2), 34: (256, 160, 1), 35: (128, 80, 2)}
CONVS = [(2, 32, 32, 1280, 1280), (2, 32, 32, 2560, 1280), (2, 32, 32, 1920, 1280), (2, 64, 64, 640, 640), (2, 64, 64, 1280, 640),
         (2, 64, 64, 1920, 640), (2, 32, 32, 640, 1280), (2, 32, 32, 128, 2560), (2, 64, 64, 128, 1280),
         (8, 32, 32, 1280, 1280), (8, 64, 64, 640, 640)]
rows = []
for (B, H, W, Cin, Cout) in CONVS:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    ws = [(torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF) for _ in range(8)]
    bias = torch.randn(Cout, device=dev)
    rb = torch.randn(B, Cout, device=dev).to(BF)
    out = torch.empty(B, H, W, Cout, device=dev, dtype=BF)
    M = B * H * W
    fl = 2.0 * M * Cout * 9 * Cin
    state = {"i": 0}

    def run(t):
        state["i"] = (state["i"] + 1) % 8
        return ops.conv3x3(x, ws[state["i"]], bias, rowbias=rb, tile=t, out=out)

    tiles = [t for t, (bm, bn, ks) in OLD.items() if M % bm == 0 and Cout % bn == 0 and Cin % (64 * ks) == 0]
    tiles += [t for t, (bm, bn, ks, hw) in HALO.items() if W == hw and (H * W) % bm == 0 and Cout % bn == 0 and Cin % (64 * ks) == 0]
    fns = {f"tile{t}": (lambda t=t: run(t)) for t in tiles}
    r = ab(fns, max(3, int(3000.0 / (fl / 1e9))))
    row = {"shape": [B, H, W, Cin, Cout], "gflop": round(fl / 1e9, 2)}
    for k, (med, mn) in r.items():
        row[k] = {"us_median": round(med, 1), "us_min": round(mn, 1), "tflops_median": round(fl / med / 1e6, 1)}
    best_old = min((row[f"tile{t}"]["us_median"], t) for t in tiles if t in OLD)
    halo = [(row[f"tile{t}"]["us_median"], t) for t in tiles if t in HALO]
    if halo:
        best_halo = min(halo)
        row["best_implicit"], row["best_halo"] = best_old[1], best_halo[1]
        row["halo_speedup"] = round(best_old[0] / best_halo[0], 3)
        state["i"] = 0
        a = ops.conv3x3(x, ws[0], bias, rowbias=rb, tile=best_old[1]).float()
        b = ops.conv3x3(x, ws[0], bias, rowbias=rb, tile=best_halo[1]).float()
        row["max_abs_diff"] = round((a - b).abs().max().item(), 5)
        row["rel_l2_diff"] = float(f"{((a - b).norm() / a.norm()).item():.3e}")
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump({"device": torch.cuda.get_device_name(0), "rounds": ROUNDS, "rows": rows}, open(out_path, "w"), indent=1)
