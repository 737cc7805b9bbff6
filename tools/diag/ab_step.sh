#!/bin/bash
# A/B of two builds of libstgraph_hip.so on one box: build/ab/{old,new}.so, step microbench at two sizes each, twice.
set -e
for round in 1 2; do
  for v in old new; do
    cp build/ab/$v.so stgraph_amd/lib/libstgraph_hip.so
    for n in 50000 49152; do
      echo -n "$v N=$n: "
      timeout -k 10 120 python tools/microbench_step.py --nodes $n --edges $((n*10)) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(v,1) for k,v in d.items() if k.startswith('step')})"
    done
  done
done
cp build/ab/new.so stgraph_amd/lib/libstgraph_hip.so
