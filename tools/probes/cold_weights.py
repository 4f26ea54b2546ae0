"""How much of the in-step time of the M = 2048 GEMMs is cold weights?  (round 6)  The (2048, 1280, 1280) / (2048, 1280, 5120) launches on tile 35
and the (2048, 10240, 1280) GEGLU on tile 37, cycling over n distinct weight matrices: n x bytes below the 256 MB Infinity Cache (every launch
finds its weights there) vs far above it (every launch streams them from HBM, as in the step: 7.7 GB of weights per step), the latter also
with the previous launch touching the next launch's weights (supir_launch_hints.next_weight, what the step does).
Usage: python tools/probes/cold_weights.py [out.json]"""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from supir_amd import _lib, ops
from supir_amd.weights import interleave_geglu

BF, dev = torch.bfloat16, "cuda"
lib = _lib.load(BF)
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cold_weights.json"
st = torch.cuda.current_stream().cuda_stream
rows = []
for (M, N, K, tile, act) in [(2048, 1280, 1280, 35, 0), (2048, 1280, 5120, 35, 0), (2048, 10240, 1280, 37, 2), (2048, 3840, 1280, -2, 0)]:
    wbytes = N * K * 2
    a = torch.randn(M, K, device=dev).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    n_out = N // 2 if act == 2 else N
    out = torch.empty(M, n_out, device=dev, dtype=BF)
    row = {"gemm": [M, N, K], "tile": tile, "weight_MB": round(wbytes / 1e6, 1)}
    for label, n in (("resident", max(2, int(64e6 // wbytes))), ("streamed", int(1.2e9 // wbytes))):
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(n)]
        if act == 2:
            ws = [interleave_geglu(w, None, 16)[0] for w in ws]

        def launch(i, hint):
            w = ws[i % n]
            h = None
            if hint:
                nx = ws[(i + 1) % n]
                h = _lib.LaunchHints(next_weight=nx.data_ptr(), next_weight_bytes=nx.numel() * 2, gn_partials_out=None)
            if tile == -2:      # the fused q|k|v projection
                return ops.gemm_qkv(a, w, None, 2, M // 2, 2560)
            rc = lib.supir_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, n_out, None, None, 0, 0,
                                        None if act == 2 else res.data_ptr(), 0 if act == 2 else N, act, 0, 1.0, tile,
                                        None if h is None else ctypes.byref(h), st)
            assert rc == 0, rc
        for hint in ((False, True) if tile != -2 else (False,)):
            for i in range(n):
                launch(i, hint)
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = max(64, n)
                e0.record()
                for i in range(iters):
                    launch(i, hint)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / iters * 1e3)
            row[f"{label}_n{n}{'_with_next_weight_hint' if hint else ''}_us"] = round(statistics.median(ts), 2)
        del ws
        torch.cuda.empty_cache()
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(out_path, "w"), indent=1)
