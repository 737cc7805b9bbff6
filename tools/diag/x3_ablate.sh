#!/bin/bash
# on the GPU box: time the split-form row product of every diagnosis build under stgraph_amd/lib/x3diag/
cd "$GRAFT_REPO_ROOT"
cp stgraph_amd/lib/libstgraph_hip.so /tmp/product.so
for f in stgraph_amd/lib/x3diag/*.so; do
  cp "$f" stgraph_amd/lib/libstgraph_hip.so
  echo "== $(basename $f)"
  timeout -k 10 120 python tools/microbench_x3.py quick 2>&1 | grep -v amdgpu.ids
done
cp /tmp/product.so stgraph_amd/lib/libstgraph_hip.so
