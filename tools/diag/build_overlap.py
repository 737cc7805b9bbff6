"""From a rocprofv3 --kernel-trace CSV of tools/diag/dyn_only.py rebuild_per_snapshot: how much of the snapshot builds' device time
(direct2_* / direct3_* launches) lies beside a launch of another queue, and which queues the launches ran on.
python tools/diag/build_overlap.py <kernel_trace.csv>"""
import csv
import json
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r["Kernel_Name"]                                     # noqa: E731
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
t_lo = ev[len(ev) // 2][0]                                            # the second half of the run: replayed epochs only
ev = [e for e in ev if e[0] >= t_lo]
build = [e for e in ev if "direct2_" in e[2] or "direct3_" in e[2]]
other = [e for e in ev if not ("direct2_" in e[2] or "direct3_" in e[2])]
queues = Counter((("build" if e in build else "other"), e[3], e[4]) for e in ev)
# time of build launches covered by some `other` launch
covered = 0
j = 0
other_iv = [(s, t) for s, t, *_ in other]
for s, t, *_ in build:
    while j < len(other_iv) and other_iv[j][1] <= s:
        j += 1
    k = j
    while k < len(other_iv) and other_iv[k][0] < t:
        covered += max(0, min(t, other_iv[k][1]) - max(s, other_iv[k][0]))
        k += 1
tot = sum(t - s for s, t, *_ in build)
span = ev[-1][1] - ev[0][0]
busy_other = sum(t - s for s, t in other_iv)
print(json.dumps({"launches": len(ev), "build_launches": len(build), "build_device_ms": tot / 1e6, "build_ms_beside_another_launch": covered / 1e6,
                  "share_beside": covered / max(tot, 1), "span_ms": span / 1e6, "other_device_ms": busy_other / 1e6,
                  "queues": {str(k): v for k, v in queues.items()}}))
# step launches beside a build launch against alone: mean duration, and the idle gap in front of them on their own queue
import bisect
b_iv = sorted((s, t) for s, t, *_ in build)
b_starts = [s for s, _ in b_iv]
def beside(s, t):
    i = bisect.bisect_left(b_starts, t)
    return any(b_iv[k][1] > s for k in range(max(0, i - 8), i))
stats = {}
prev_end = {}
for s, t, nm, q, st in ev:
    if (s, t, nm, q, st) in build:
        continue
    key = ("step_fwd" if "tgcn_step_fwd" in nm else "step_bwd" if "tgcn_step_bwd" in nm else "other") + ("/beside" if beside(s, t) else "/alone")
    d = stats.setdefault(key, [0, 0, 0])
    d[0] += 1
    d[1] += t - s
    if q in prev_end:
        d[2] += max(0, s - prev_end[q])
    prev_end[q] = t
print(json.dumps({k: {"n": v[0], "mean_us": v[1] / v[0] / 1e3, "mean_gap_before_us": v[2] / v[0] / 1e3} for k, v in sorted(stats.items())}))
