"""What a GEMM tile is worth INSIDE a step: throughput with two independent launch chains in flight (two streams), next to the
isolated back-to-back latency the autotuner times.  Per shape and tile: us per problem (a) serial on one stream, (b) two streams
each issuing the same launch sequence, (c) grouped two-problem launches on one stream, (d) grouped launches on two streams.
Usage: python tools/dual_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops

dev = "cuda"
BF = torch.bfloat16
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(fn_main, fn_side, n=20):
    for _ in range(2):
        fn_main()
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if fn_side is not None:
            side.wait_stream(main)
        for _ in range(n):
            fn_main()
        if fn_side is not None:
            with torch.cuda.stream(side):
                for _ in range(n):
                    fn_side()
            main.wait_stream(side)
        e1.record()
        e1.synchronize()
        dt = e0.elapsed_time(e1) * 1e3 / n
        best = dt if best is None or dt < best else best
    return best


SHAPES = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (8192, 640, 640), (8192, 640, 2560)]
for (M, N, K) in SHAPES:
    probs = []
    for s in range(4):
        g = torch.Generator().manual_seed(s)
        probs.append(((torch.randn(M, K, generator=g)).to(dev).to(BF), (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(BF),
                      torch.empty(M, N, dtype=BF, device=dev)))
    fl = 2.0 * M * N * K
    print(f"--- ({M}, {N}, {K}): us per problem [TF/s]   serial | two streams | grouped x2 | grouped x2 on two streams")
    cands = ops._gemm_candidates(M, N, K, 0, 0, N)
    for t in cands:
        def one(i, t=t):
            a, w, o = probs[i]
            return ops.gemm(a, w, None, out=o, tile=t)
        try:
            one(0)
        except Exception as e:  # noqa: BLE001
            continue
        s1 = timed(lambda: one(0), None)
        s2 = timed(lambda: one(0), lambda: one(1)) / 2
        line = f"  tile {t:2d}: {s1:7.1f} [{fl / s1 / 1e6:6.0f}] | {s2:7.1f} [{fl / s2 / 1e6:6.0f}]"
        if t in ops.PAIR_TILES:
            def grp(i, j, t=t):
                return ops.paired_run(lambda: one(i), lambda: one(j))
            tr = ops.start_trace()
            grp(0, 1)
            ops.stop_trace()
            if any(r.get("group") == 2 for r in tr):
                g1 = timed(lambda: grp(0, 1), None) / 2
                g2 = timed(lambda: grp(0, 1), lambda: grp(2, 3)) / 4
                line += f" | {g1:7.1f} [{fl / g1 / 1e6:6.0f}] | {g2:7.1f} [{fl / g2 / 1e6:6.0f}]"
        print(line, flush=True)
