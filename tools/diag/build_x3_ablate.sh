#!/bin/bash
# Diagnosis builds of the split-form row product with one ingredient removed (-DSTG_X3_ABLATE=bits: 1 no stores, 2 every load
# from one cached tile, 4 no matrix instructions) -> stgraph_amd/lib/x3diag/abl<bits>.so.  Wrong results by construction:
# tools/diag/x3_ablate.sh copies each over the product library on the GPU box and times tools/microbench_x3.py.
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
OBJS=$(ls ../../build/obj/*.o | grep -v rowgemm_x3)
mkdir -p ../lib/x3diag ../../build/x3diag
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DSTG_X3_ABLATE=$bits -c rowgemm_x3.hip -o ../../build/x3diag/abl$bits.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/x3diag/abl$bits.so $OBJS ../../build/x3diag/abl$bits.o -lhiprtc
done
ls -la ../lib/x3diag/
