#!/bin/bash
# Diagnosis builds of gemm_tn_x3.hip (WRONG results by construction; loaded through STGRAPH_AMD_LIB only):
#   stgraph_amd/lib/diag/x3g_noproducts.so (loads only), x3g_noloads.so (splits + products only)
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
mkdir -p ../../build/x3g ../lib/diag
for v in 1 2; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 --offload-compress -fPIC -ffp-contract=off -std=c++17 -DSTG_X3G_ABLATE=$v -c gemm_tn_x3.hip -o ../../build/x3g/gemm_tn_x3_$v.o
  OBJS=$(ls ../../build/obj/*.o | grep -v gemm_tn_x3)
  name=$([ $v = 1 ] && echo x3g_noproducts || echo x3g_noloads)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/diag/$name.so $OBJS ../../build/x3g/gemm_tn_x3_$v.o -lhiprtc
done
ls -la ../lib/diag/x3g_*.so
