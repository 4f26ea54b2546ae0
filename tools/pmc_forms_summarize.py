"""Per-kernel averages of the rocprofv3 --pmc CSVs of tools/pmc_conv_forms.py: python tools/pmc_forms_summarize.py out.json a.csv b.csv ..."""
import collections
import csv
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            if not name.startswith("gemm16_kernel"):
                continue
            name = re.sub(r"\(.*$", "", name)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, cs in agg.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_LDS_IDX_ACTIVE" in e and e["SQ_LDS_IDX_ACTIVE"]:
        e["lds_bank_conflict_over_idx_active"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    if "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS"):
            if c in e:
                e[c + "_over_SQ_BUSY_CYCLES"] = e[c] / e["SQ_BUSY_CYCLES"]
    res[k] = e
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, e in res.items():
    print(k)
    print("   ", {c: (round(v, 4) if v < 100 else round(v)) for c, v in e.items()})
