#!/bin/bash
# usage: tools/pmc_gcn.sh <tag>     -> gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum_TCC_MISS_sum}/...
# The three SEPARATE rocprofv3 --pmc passes (counters only, no trace domains) over tools/pmc_gcn.py that
# tools/summarize_profiles.py pmc reduces, plus FETCH_SIZE / WRITE_SIZE over tools/pmc_cora.py (narrow rows).
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
dst=gpurun_out/pmc_$tag
mkdir -p "$dst"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  name=${c// /_}
  out=/tmp/pmc_${tag}_$name
  rm -rf "$out"
  echo "pass $name (cfg2)"
  rocprofv3 --pmc $c --output-format csv -d "$out" -- python3 tools/pmc_gcn.py --iters 3 > "$dst/$name.json" 2> "$dst/$name.stderr" || echo "pass $name failed"
  mkdir -p "$dst/$name"
  find "$out" -name "*counter_collection.csv" -exec cp {} "$dst/$name/" \;
done
for c in FETCH_SIZE WRITE_SIZE; do
  out=/tmp/pmc_${tag}_cora_$c
  rm -rf "$out"
  echo "pass $c (cora x 1024)"
  rocprofv3 --pmc $c --output-format csv -d "$out" -- python3 tools/pmc_cora.py --iters 3 > "$dst/cora_$c.json" 2> "$dst/cora_$c.stderr" || echo "pass cora $c failed"
  mkdir -p "$dst/cora_$c"
  find "$out" -name "*counter_collection.csv" -exec cp {} "$dst/cora_$c/" \;
done
ls -la "$dst"
