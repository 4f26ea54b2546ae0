"""Interleaved A/B (round 5): (1) the GEGLU projections on tile 37 (256 x 320, csrc/gemm_big.hip), 34 (256 x 160) and 44 (256 x 320 on the
eight-phase schedule) at the CFG batch of one image and of tile batches; (2) supir_xattn_q_d64 with the 2-D XCD grid vs the 1-D ranges
(tools knob 5).  Usage: python tools/bench_geglu_xattn.py out.json"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import _lib, ops
from supir_amd.weights import interleave_geglu

BF, dev = torch.bfloat16, "cuda"
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/geglu_xattn.json"
ROUNDS = 7


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


rows = []
for (M, K, N2) in [(2048, 1280, 10240), (8192, 1280, 10240), (8192, 640, 5120), (32768, 640, 5120)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N2, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N2, device=dev)
    w16, b16 = interleave_geglu(w, b, 16)
    fl = 2.0 * M * N2 * K
    fns = {f"tile{t}": (lambda t=t: ops.gemm(a, w16, b16, act=2, tile=t)) for t in (37, 34, 44)}
    r = ab(fns, max(3, int(2000.0 / (fl / 1e9))))
    row = {"kind": "geglu", "shape": [M, N2, K]}
    for k, (med, mn) in r.items():
        row[k] = {"us_median": round(med, 1), "us_min": round(mn, 1), "tflops_median": round(fl / med / 1e6, 1)}
    row["tile44_bitwise_tile34"] = bool(torch.equal(ops.gemm(a, w16, b16, act=2, tile=44), ops.gemm(a, w16, b16, act=2, tile=34)))
    rows.append(row)
    print(json.dumps(row), flush=True)
lib = _lib.load()
for (B, H, T, Tk, C) in [(2, 20, 1024, 77, 1280), (2, 10, 4096, 77, 640), (8, 20, 1024, 77, 1280)]:
    N = H * 64
    x = torch.randn(B, T, C, device=dev).to(BF)
    wq = (torch.randn(N, C, device=dev) * C ** -0.5).to(BF)
    k = torch.randn(B, Tk, N, device=dev).to(BF)
    vt = torch.zeros(B, N, 128, device=dev, dtype=BF)
    vt[:, :, :Tk] = torch.randn(B, N, Tk, device=dev).to(BF)
    fl = 2.0 * B * T * N * C + 4.0 * B * T * N * Tk
    # cold-ish weights: rotate over 16 copies of W_q (53 MB at C = 1280) so that the launch does not find its weight rows in L2
    wqs = [wq.clone() for _ in range(16)]
    state = {"i": 0}

    def run(knob):
        state["i"] = (state["i"] + 1) % 16
        with _lib.tools_knob(5, knob):      # both arms on libsupir_hip_tools.so (the only build with variant switches)
            return ops.xattn_q(x, wqs[state["i"]], None, k, vt, B, H, T, Tk)

    r = ab({"grid2d": lambda: run(0), "ranges1d": lambda: run(1)}, 40)
    row = {"kind": "xattn_q", "shape": [B, H, T, Tk, C]}
    for kk, (med, mn) in r.items():
        row[kk] = {"us_median": round(med, 2), "us_min": round(mn, 2), "tflops_median": round(fl / med / 1e6, 1)}
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(rows, open(out_path, "w"), indent=1)
