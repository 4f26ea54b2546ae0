"""Per-kernel averages of rocprofv3 --pmc counter CSVs (one or more files / counters): counter value per launch, grouped by
(short kernel name, grid size).  FETCH_SIZE / WRITE_SIZE are reported in bytes (the counters are KiB); FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies 128-byte requests at 64 B for wide coalesced streams).
Usage: python tools/pmc_summarize.py out.json file1.csv [file2.csv ...]"""
import collections
import csv
import json
import re
import sys

out_path, files = sys.argv[1], sys.argv[2:]
PER_CASE = 3    # tools/pmc_probe.py launches every case three times, cases in a fixed order: two cases that share an instantiation AND a
                # grid size (K = 1280 and K = 5120 on the 128 x 80 tile) are told apart by their position in the dispatch order ("#0", "#1")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    rows = collections.defaultdict(list)
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", ""))
            if not (name.startswith("gemm") or name.startswith("geglu") or name.startswith("attn") or name.startswith("xattn") or name.startswith("gn_")):
                continue
            rows[f"{name} grid={r['Grid_Size']}"].append(r)
    for base, rs in rows.items():
        ids = sorted({int(r["Dispatch_Id"]) for r in rs})
        block = {d: i // PER_CASE for i, d in enumerate(ids)}
        nblocks = max(block.values()) + 1
        for r in rs:
            key = base if nblocks == 1 else f"{base} #{block[int(r['Dispatch_Id'])]}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for key, cs in sorted(agg.items()):
    e = {"launches": len(cs["_dur_us"]) // max(1, len([c for c in cs if not c.startswith("_")]))}
    for c, vals in cs.items():
        v = sum(vals) / len(vals)
        if c == "FETCH_SIZE":
            e["fetch_bytes"] = v * 1024 * 2
        elif c == "WRITE_SIZE":
            e["write_bytes"] = v * 1024
        elif c == "_dur_us":
            e["profiled_launch_us"] = v
        else:
            e[c] = v
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
        e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
    res[key] = e
json.dump(res, open(out_path, "w"), indent=1)
for k, e in res.items():
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in e.items()})
