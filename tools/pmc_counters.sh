#!/bin/bash
# usage: tools/pmc_counters.sh <tag> <kernel-name-substring> "<counters of pass 0>" "<counters of pass 1>" ... -- <python script + args...>
# One rocprofv3 --pmc pass (counters only, no trace domains) per quoted counter list, each under its own timeout;
# per-dispatch means of the matching kernels -> gpurun_out/pmc_<tag>.json
tag=$1; shift
match=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for p in "${passes[@]}"; do
  out=/tmp/pmc_${tag}_$i
  rm -rf "$out"
  echo "pass $i: $p"
  timeout -k 10 150 rocprofv3 --pmc $p --output-format csv -d "$out" -- python3 "$@" > gpurun_out/pmc_${tag}_$i.stdout 2> gpurun_out/pmc_${tag}_$i.stderr || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.stderr; }
  find "$out" -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_${tag}_$i.csv \;
  i=$((i+1))
done
python3 - "$tag" "$match" <<'PY'
import csv, glob, json, sys, collections
tag, match = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"][:100] + " grid=" + r.get("Grid_Size", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k, d in per.items():
        n = len(cnt[k])
        res.setdefault(k, {"dispatches": n})
        for c, v in d.items():
            res[k][c] = v / n
json.dump(res, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
