"""Within-process interleaved A/B of the large-M tiles (round 5): the VAE's 3x3 convolutions at the sizes of a 1024^2 image and the
M >= 8192 GEMMs of tile batches / `--num_samples`, on tiles 40 (256 x 256, one barrier per K step), 42 (256 x 256, eight-phase
ping-pong schedule), 39, 34 and gemm.hip's 5.  Rounds are interleaved (tile A, tile B, ... repeated) so that clock drift hits every
variant alike; reports median and min microseconds and TFLOP/s on random data, plus bitwise 42 == 40.
Usage: python tools/bench_big_tiles.py out.json [--rounds 7]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

BF, dev = torch.bfloat16, "cuda"
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/big_tiles.json"
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 7


def ab(fns, iters):
    """fns: {name: callable}; returns {name: (median_us, min_us)} over ROUNDS interleaved rounds of `iters` launches."""
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


rows = []
CONVS = [(1, 256, 256, 512, 512, False), (1, 128, 128, 512, 512, True), (1, 512, 512, 256, 256, False), (1, 256, 256, 256, 256, True),
         (1, 256, 256, 256, 512, False), (1, 256, 256, 512, 256, False), (1, 512, 512, 128, 256, False),
         (1, 1024, 1024, 128, 128, False), (1, 512, 512, 256, 128, False), (1, 512, 512, 256, 256, True), (1, 1024, 1024, 256, 128, False)]
for (B, H, W, Cin, Cout, up) in CONVS:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    bias = torch.randn(Cout, device=dev)
    OH, OW = (2 * H, 2 * W) if up else (H, W)
    M = B * OH * OW
    fl = 2.0 * M * Cout * 9 * Cin
    tiles = [t for t in (40, 42, 39, 45, 5) if not (t in (40, 42) and Cout % 256) and not (t in (39, 45) and Cout % 128)
             and not (t == 42 and (OH * OW) % 256) and not (t == 45 and (OH * OW) % 512)]
    fns = {f"tile{t}": (lambda t=t: ops.conv3x3(x, w, bias, upsample=up, tile=t)) for t in tiles}
    if 42 in tiles:      # tile 42 with the tap-major K order of every other tile (tools knob 6 = 1) beside its chunk-major default
        from supir_amd import _lib

        def tapmajor():
            with _lib.tools_knob(6, 1):      # libsupir_hip_tools.so: the only build with variant switches
                return ops.conv3x3(x, w, bias, upsample=up, tile=42)
        fns["tile42_tapmajor"] = tapmajor
    r = ab(fns, max(3, int(2000.0 / (fl / 1e9))))   # fl / 1e9 = microseconds at 1 PFLOP/s: ~2 ms of launches per round
    row = {"kind": "conv3x3", "shape": [B, H, W, Cin, Cout, int(up)], "M": M}
    for k, (med, mn) in r.items():
        row[k] = {"us_median": round(med, 1), "us_min": round(mn, 1), "tflops_median": round(fl / med / 1e6, 1)}
    if 40 in tiles and 42 in tiles:
        row["tile42_tapmajor_bitwise_tile40"] = bool(torch.equal(fns["tile42_tapmajor"](), ops.conv3x3(x, w, bias, upsample=up, tile=40)))
        a_, b_ = ops.conv3x3(x, w, bias, upsample=up, tile=42).float(), ops.conv3x3(x, w, bias, upsample=up, tile=40).float()
        row["tile42_rel_l2_vs_tile40"] = float(((a_ - b_).norm() / b_.norm()).item())
    if 39 in tiles and 45 in tiles:
        a_, b_ = ops.conv3x3(x, w, bias, upsample=up, tile=45).float(), ops.conv3x3(x, w, bias, upsample=up, tile=39).float()
        row["tile45_rel_l2_vs_tile39"] = float(((a_ - b_).norm() / b_.norm()).item())
    rows.append(row)
    print(json.dumps(row), flush=True)
    del x, w
    torch.cuda.empty_cache()
GEMMS = [(8192, 8192, 8192), (4096, 4096, 4096), (1048576, 128, 256)]
for (M, N, K) in GEMMS:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    fl = 2.0 * M * N * K
    tiles = [t for t in (40, 42, 39, 45, 5) if not (t in (40, 42) and N % 256) and not (t in (39, 45) and N % 128) and not (t == 45 and M % 512)
             and not (t == 33 and K % 128)]
    fns = {f"tile{t}": (lambda t=t: ops.gemm(a, w, None, residual=res, tile=t)) for t in tiles}
    r = ab(fns, max(3, int(2000.0 / (fl / 1e9))))
    row = {"kind": "gemm", "shape": [M, N, K]}
    for k, (med, mn) in r.items():
        row[k] = {"us_median": round(med, 1), "us_min": round(mn, 1), "tflops_median": round(fl / med / 1e6, 1)}
    if 40 in tiles and 42 in tiles:
        row["tile42_bitwise_tile40"] = bool(torch.equal(ops.gemm(a, w, None, residual=res, tile=42), ops.gemm(a, w, None, residual=res, tile=40)))
    if 39 in tiles and 45 in tiles:
        row["tile45_bitwise_tile39"] = bool(torch.equal(ops.gemm(a, w, None, residual=res, tile=45), ops.gemm(a, w, None, residual=res, tile=39)))
    rows.append(row)
    print(json.dumps(row), flush=True)
    del a, w, res
    torch.cuda.empty_cache()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(rows, open(out_path, "w"), indent=1)
