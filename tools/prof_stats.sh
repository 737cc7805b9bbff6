#!/bin/bash
# usage: tools/prof_stats.sh <tag> <python script + args...>
# rocprofv3 --kernel-trace --stats into /tmp, keep only the (small) stats CSV under gpurun_out/.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=/tmp/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$@" > gpurun_out/prof_$tag.stdout 2> gpurun_out/prof_$tag.stderr
mkdir -p gpurun_out/prof_$tag
find "$out" -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$tag/kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$tag/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); n=sum(int(r["Calls"]) for r in rows)
print("total kernel ms %.1f calls %d" % (tot/1e6, n))
for r in rows[:40]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
