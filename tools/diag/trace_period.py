#!/usr/bin/env python3
"""Kernels (and memory copies, if traced) of the last period of a rocprofv3 trace, in order, with durations and gaps.
argv: <dir/prefix> marker   (reads <prefix>_kernel_trace.csv and, when present, <prefix>_memory_copy_trace.csv)"""
import csv, os, sys
pre = sys.argv[1]
rows = [dict(r, what=r["Kernel_Name"]) for r in csv.DictReader(open(pre + "_kernel_trace.csv"))]
mc = pre + "_memory_copy_trace.csv"
if os.path.exists(mc):
    for r in csv.DictReader(open(mc)):
        rows.append(dict(r, what="MEMCPY %s %s bytes" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "?")))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["what"]]
a, b = idx[-2], idx[-1]
per = rows[a:b]
t0 = int(per[0]["Start_Timestamp"])
prev = None
for r in per:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f us  dur %6.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["what"][:100]))
    prev = e
print("period us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "entries", len(per))
