#!/bin/bash
# usage: tools/pmc_mfma.sh <tag> <kernel-name-substring> <python script + args...>
# One rocprofv3 --pmc pass with the matrix-core counters -> gpurun_out/pmc_<tag>_mfma.csv (per-dispatch values).
tag=$1; shift
match=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=/tmp/pmc_${tag}_mfma
rm -rf "$out"
echo "pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out" -- python3 "$@" > gpurun_out/pmc_${tag}_mfma.stdout 2> gpurun_out/pmc_${tag}_mfma.stderr || echo "pass failed"
find "$out" -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_${tag}_mfma.csv \;
python3 - "$tag" "$match" <<'PY'
import csv, sys, collections
tag, match = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"gpurun_out/pmc_{tag}_mfma.csv")):
    if match in r["Kernel_Name"]:
        per[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in per.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
