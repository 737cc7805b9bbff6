"""The launches of ONE captured Cora epoch in order, with device time and the idle gap in front of each: from a rocprofv3 --kernel-trace
CSV of `python tools/diag/cora_only.py` (the last complete replay).  python tools/diag/cora_kernels.py <kernel_trace.csv>"""
import csv
import json
import sys

rows = sorted(list(csv.DictReader(open(sys.argv[1]))), key=lambda r: int(r["Start_Timestamp"]))
# replays are separated by the host's sync + timing code: a gap > 20 us between launches
groups, cur, last = [], [], None
for r in rows:
    s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last is not None and s - last > 20_000:
        groups.append(cur)
        cur = []
    cur.append((r["Kernel_Name"][:100], s, t))
    last = t
groups.append(cur)
pick = int(sys.argv[2]) if len(sys.argv) > 2 else -2                     # which group (default: the last complete replay)
g = groups[pick]
out, prev = [], None
for nm, s, t in g:
    out.append({"kernel": nm, "us": (t - s) / 1e3, "gap_us": 0.0 if prev is None else (s - prev) / 1e3})
    prev = t
print(json.dumps({"launches": len(g), "span_us": (g[-1][2] - g[0][1]) / 1e3, "device_us": sum(o["us"] for o in out),
                  "sizes_of_last_groups": [len(x) for x in groups[-6:]], "kernels": out}, indent=1))
