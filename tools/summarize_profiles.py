#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the small files committed under profiles/.

  python tools/summarize_profiles.py stats <dir with *_kernel_stats.csv> <out.csv>   (top kernels)
  python tools/summarize_profiles.py pmc <gpurun_out/pmc_rNN> <out.json>            (gcn_agg traffic)

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come from SEPARATE passes, are in KiB, WRITE_SIZE is exact for 16-B-per-lane stores,
and FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 -- the factor is measured here
on a launch with a known byte count and the same access shape (tools/pmc_gcn.py, launch A)
instead of being assumed.
"""
import csv
import glob
import json
import sys


def stats(src, dst, top=25):
    f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(dst, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:top]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    print("wrote", dst)


def _counter(d, name):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    # the row-group kernel only: on large narrow-row graphs a (nearly empty) long-row launch follows each call
    rows = [r for r in csv.DictReader(open(f)) if "gcn_agg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6) for r in rows]


def pmc(src, dst):
    meta = json.loads(open(f"{src}/FETCH_SIZE.json").read().strip().splitlines()[-1])
    it = meta["iters"]
    fetch = _counter(f"{src}/FETCH_SIZE", "FETCH_SIZE")
    write = _counter(f"{src}/WRITE_SIZE", "WRITE_SIZE")
    hit = _counter(f"{src}/TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum")
    miss = _counter(f"{src}/TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum")
    mean = lambda xs: sum(xs) / len(xs)  # noqa: E731
    seg = lambda rows, k: [v for v, _ in rows[k * it:(k + 1) * it]]  # noqa: E731
    cal_fetch = mean(seg(fetch, 0)) * 1024
    cal_write = mean(seg(write, 0)) * 1024
    known_r, known_w = meta["calibration"]["known_read_bytes"], meta["calibration"]["known_write_bytes"]
    corr = known_r / cal_fetch
    out = {"source": src, "unit": "bytes per launch",
           "calibration": {"launch": "gcn_agg on an identity graph, |V|=4M, F=128 (every row read once)",
                           "known_read_bytes": known_r, "FETCH_SIZE_bytes_raw": cal_fetch,
                           "fetch_correction_factor": corr, "guide_factor": 2.0,
                           "known_write_bytes": known_w, "WRITE_SIZE_bytes_raw": cal_write,
                           "write_ratio": cal_write / known_w}}
    for k, name in ((1, "cfg2_forward_csr"), (2, "cfg2_backward_csr")):
        fr, wr = mean(seg(fetch, k)) * 1024, mean(seg(write, k)) * 1024
        h, m = mean(seg(hit, k)), mean(seg(miss, k))
        dur_ms = mean([t for _, t in fetch[k * it:(k + 1) * it]])
        out[name] = {"FETCH_SIZE_bytes_raw": fr, "WRITE_SIZE_bytes": wr,
                     "read_bytes_corrected_measured_factor": fr * corr, "read_bytes_corrected_x2": fr * 2,
                     "traffic_bytes": fr * corr + wr, "traffic_bytes_x2_rule": fr * 2 + wr,
                     "algorithmic_bytes": meta["cfg2"]["algorithmic_bytes"],
                     "compulsory_bytes": meta["cfg2"]["compulsory_bytes"],
                     "traffic_over_algorithmic": (fr * corr + wr) / meta["cfg2"]["algorithmic_bytes"],
                     "L2_hit_rate": h / (h + m), "kernel_ms_under_profiler": dur_ms}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
