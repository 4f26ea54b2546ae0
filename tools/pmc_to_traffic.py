"""Merge a pmc_summary.json (tools/pmc_summarize.py) into profiles/pmc_traffic.json: per bench kernel name and shape the
counter-measured fetch / write bytes per launch, their ratio to the algorithmic bytes, and (when the SQ pass is present) MFMA busy
over SQ busy and the LDS bank-conflict ratio.  Cases (tools/pmc_probe.py's cases.json) are matched to counter rows by kernel family and
grid size (work-items = workgroups x threads).
Usage: python tools/pmc_to_traffic.py cases.json pmc_summary.json profiles/pmc_traffic.json "note text" """
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supir_amd.ops import gemm_tile_name  # noqa: E402

cases, summ, out_path, note = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3], sys.argv[4]
G16 = {32: (128, 80), 33: (128, 160), 34: (256, 160), 35: (128, 80), 37: (256, 320), 39: (256, 128), 40: (256, 256), 42: (256, 256), 45: (512, 128)}
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
out["note"] = note


seen = {}


def find(prefix, grid):
    """The counter row of the next case on this (kernel, grid): cases that share both come in dispatch order ("#0", "#1", ...)."""
    keys = sorted(k for k in summ if k.startswith(prefix) and (k.endswith(f"grid={grid}") or f"grid={grid} #" in k))
    if not keys:
        return None
    i = seen.get((prefix, grid), 0)
    seen[(prefix, grid)] = i + 1
    return summ[keys[min(i, len(keys) - 1)]]


for c in cases:
    kind, shape, tile = c["kind"], c["shape"], c["tile"]
    if kind in ("gemm", "gemm_geglu", "conv3x3"):
        if kind == "conv3x3":
            b, hw, ch = shape.split()
            B, (H, W), (Cin, Cout) = int(b[1:]), map(int, hw.split("x")), map(int, ch.split("->"))
            M, N = B * H * W, Cout
        else:
            M, N = (int(t[1:]) for t in shape.split()[:2])
        bm, bn = G16[tile]
        grid = (M // bm) * (N // bn) * 512
        targs = {32: "128, 80, 4, 1, 2, 2", 33: "128, 160, 2, 2, 2, 2", 34: "256, 160, 8, 1, 1, 3", 35: "128, 80, 4, 1, 2, 3",
                 39: "256, 128, 4, 2, 1, 3", 40: "256, 256, 4, 2, 1, 2", 42: "256, 256, 2, 4, 1, 8", 45: "512, 128, 4, 2, 1, 8"}
        if tile == 34 and (kind == "conv3x3" or (kind == "gemm" and M >= 8192)):
            targs[34] = "256, 160, 4, 2, 1, 3"     # round 4: the eight waves as 4 x 2 for convolutions and M >= 8192
        e = find("geglu_big_kernel", grid) if tile == 37 else \
            find(f"gemm16_kernel<{targs[tile]}, false, {'true' if kind == 'conv3x3' else 'false'}, false, 1>", grid)
        name = gemm_tile_name(M, N, 2 if kind == "gemm_geglu" else 0, conv=(kind == "conv3x3"), tile=tile)
    elif kind == "gemm_qkv":      # fused q|k|v projection on 256 x 128 tiles (round 4)
        M, N = (int(t[1:]) for t in shape.split()[:2])
        e = find("gemm16_kernel<256, 128, 4, 2, 1, 3, false, false, true, 1>", (M // 256) * (N // 128) * 512)
        name = gemm_tile_name(M, N, 0, tile=tile)
    elif kind == "attn":
        B, H, Tq = (int(t.lstrip("BHTq")) for t in shape.split()[:3])
        e = find("attn_d64_pipe_kernel", ((Tq + 127) // 128) * H * B * 256)
        name = "attn"
    elif kind == "xattn_q":
        B, H, T = (int(t.lstrip("BHT")) for t in shape.split()[:3])
        e = find("xattn_q_kernel", (T // 128) * H * B * 256)
        name = "xattn_q"
    else:
        continue
    if e is None:
        print("no counter row for", kind, shape, tile)
        continue
    rec = {"shape": shape + (" geglu" if kind == "gemm_geglu" else ""), "algorithmic_bytes": c["algorithmic_bytes"]}
    if "fetch_bytes" in e:
        rec["fetch_bytes"] = e["fetch_bytes"]
    if "write_bytes" in e:
        rec["write_bytes"] = e["write_bytes"]
    if "fetch_bytes" in e and "write_bytes" in e:
        rec["fetch_plus_write_over_algorithmic"] = round((e["fetch_bytes"] + e["write_bytes"]) / c["algorithmic_bytes"], 2)
    if "mfma_busy_over_sq_busy" in e:
        # SQ_VALU_MFMA_BUSY_CYCLES accumulates over the 32 SIMD-slots an SQ_BUSY_CYCLES tick spans (profiles/r02: 8.8 "of 32"):
        # / 32 = the share of cycles the matrix pipes were busy while the kernel was resident
        rec["mfma_busy_over_sq_busy"] = round(e["mfma_busy_over_sq_busy"], 3)
        rec["mfma_busy_frac"] = round(e["mfma_busy_over_sq_busy"] / 32.0, 3)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_over_idx_active"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)
    if e.get("SQ_WAVE_CYCLES"):      # shares of the resident wave-cycles (SQ_WAVE_CYCLES counts quad-cycles, like the SQ_WAIT_* / SQ_ACTIVE_* counters)
        for cname, key in (("SQ_WAIT_ANY", "wait_any_over_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_over_wave_cycles"),
                           ("SQ_ACTIVE_INST_ANY", "active_inst_any_over_wave_cycles"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_over_wave_cycles")):
            if cname in e:
                rec[key] = round(e[cname] / e["SQ_WAVE_CYCLES"], 3)
    if e.get("GRBM_GUI_ACTIVE") and e.get("profiled_launch_us"):
        # GRBM_GUI_ACTIVE = cycles the graphics engine was busy during the dispatch: / its duration = the effective shader clock under this
        # kernel's load (MI355X_MICROARCH.md, DVFS give-back); summed over the 8 XCDs' instances by rocprofv3 -> / 8
        rec["effective_clock_ghz"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / e["profiled_launch_us"] / 1e3, 3)
        rec["profiled_launch_us"] = round(e["profiled_launch_us"], 1)
    lst = [x for x in out.get(name, []) if isinstance(x, dict) and x.get("shape") != rec["shape"]] if isinstance(out.get(name), list) else []
    out[name] = lst + [rec]
# GroupNorm apply kernel: conflict ratio per grid (no traffic model needed: read + write once)
gn = {k: {"lds_bank_conflict_over_idx_active": round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)}
      for k, e in summ.items() if k.startswith("gn_apply_kernel") and e.get("SQ_LDS_IDX_ACTIVE")}
if gn:
    out["groupnorm_apply_lds"] = gn
json.dump(out, open(out_path, "w"), indent=1)
print("updated", out_path, "kernels:", [k for k in out if k != "note"])
