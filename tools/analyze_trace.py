"""Cut a rocprofv3 kernel-trace CSV into idle-separated segments and summarise the LAST isolated graph replay of the step:
wall time, per-queue busy time and gaps, concurrency, and per-kernel-name in-step durations.
Usage: python tools/analyze_trace.py <kernel_trace.csv> [out.json]"""
import collections
import csv
import json
import re
import sys

path = sys.argv[1]
rows = []
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                     int(r.get("Workgroup_Size_X", 0) or 0), int(r.get("Grid_Size_X", 0) or 0)))
rows.sort()
# segments: split where the GPU idles for > 50 ms
segs, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[0] - last_end > 50e6:
        segs.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end or 0, r[1])
segs.append(cur)
print("segments (kernels):", [len(s) for s in segs])
cand = [s for s in segs if 900 <= len(s) <= 1500]
seg = cand[-1] if cand else max(segs, key=len)
t0, t1 = seg[0][0], max(r[1] for r in seg)
wall = (t1 - t0) / 1e3
print(f"step segment: {len(seg)} kernels, wall {wall / 1e3:.3f} ms")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


by_q = collections.defaultdict(list)
for r in seg:
    by_q[r[3]].append(r)
out = {"kernels": len(seg), "wall_us": wall, "queues": {}}
for q, rs in by_q.items():
    rs.sort()
    busy = sum(r[1] - r[0] for r in rs) / 1e3
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    span = (rs[-1][1] - rs[0][0]) / 1e3
    out["queues"][q] = dict(kernels=len(rs), busy_us=busy, span_us=span, first_us=(rs[0][0] - t0) / 1e3, last_us=(rs[-1][1] - t0) / 1e3,
                            gap_sum_us=sum(pos), gap_median_us=sorted(pos)[len(pos) // 2] if pos else 0,
                            gaps_over_5us=sum(1 for g in pos if g > 5), gaps_over_20us=sum(1 for g in pos if g > 20))
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.2f} ms, span {span / 1e3:.2f} ms [{(rs[0][0] - t0) / 1e6:.2f}..{(rs[-1][1] - t0) / 1e6:.2f}], "
          f"gaps sum {sum(pos) / 1e3:.2f} ms median {out['queues'][q]['gap_median_us']:.2f} us, >5us: {out['queues'][q]['gaps_over_5us']}, >20us: {out['queues'][q]['gaps_over_20us']}")
# concurrency profile: time with 0 / 1 / 2+ kernels in flight
ev = []
for r in seg:
    ev.append((r[0], 1))
    ev.append((r[1], -1))
ev.sort()
lvl, prev, hist = 0, t0, collections.Counter()
for ts, d in ev:
    hist[min(lvl, 3)] += ts - prev
    prev = ts
    lvl += d
print("time with N kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
out["in_flight_ms"] = {str(k): v / 1e6 for k, v in hist.items()}
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    a = agg[short(r[2])]
    a[0] += 1
    a[1] += (r[1] - r[0]) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"sum of kernel durations {tot / 1e3:.2f} ms")
out["sum_kernel_us"] = tot
out["by_kernel"] = {}
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {us / 1e3:8.3f} ms  {n:5d} x {us / n:8.2f} us  {k}")
    out["by_kernel"][k] = dict(launches=n, us=us, avg_us=us / n)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
