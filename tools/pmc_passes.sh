#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <kernel-name-substring> <python script + args...>
# Separate rocprofv3 --pmc passes (counters only: no trace domains), one CSV per pass, reduced to per-dispatch
# sums for the kernels whose name contains the substring -> gpurun_out/pmc_<tag>.json
# (a fourth pass -- first with TA_* counters, then with six TCC_* counters -- aborted rocprofv3 with signal 6 on this
# pool and the call then sat until it was killed: only the three SQ passes are run; TCC hit / miss come from
# tools/pmc_gcn.sh, two counters per pass)
tag=$1; shift
match=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_CYCLES"
)
i=0
for p in "${passes[@]}"; do
  out=/tmp/pmc_${tag}_$i
  rm -rf "$out"
  echo "pass $i: $p"
  rocprofv3 --pmc $p --output-format csv -d "$out" -- python3 "$@" > gpurun_out/pmc_${tag}_$i.stdout 2> gpurun_out/pmc_${tag}_$i.stderr || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_$i.stderr; }
  find "$out" -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_${tag}_$i.csv \;
  i=$((i+1))
done
python3 - "$tag" "$match" <<'PY'
import csv, glob, json, sys, collections
tag, match = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum over dispatches
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"][:100]
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
