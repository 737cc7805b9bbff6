#!/bin/bash
# on the GPU box: time the split-form row product of every diagnosis build under stgraph_amd/lib/x3diag/.
# The diagnosis builds compute WRONG products by construction: they are loaded through STGRAPH_AMD_LIB, the product
# library stgraph_amd/lib/libstgraph_hip.so is never touched.
cd "$GRAFT_REPO_ROOT"
for f in stgraph_amd/lib/x3diag/*.so; do
  echo "== $(basename $f)"
  STGRAPH_AMD_LIB="$PWD/$f" timeout -k 10 120 python tools/microbench_x3.py quick 2>&1 | grep -v amdgpu.ids
done
