#!/bin/bash
# Diagnosis builds of the matrix-core TGCN step kernels with timestamps (-DSTG_STEPX_TRACE) and one ingredient removed:
#   build/tracex_nostores/libstgraph_hip.so   no global stores of row pieces (-DSTG_ABLATE_STORES: tgcn_step.hpp st_f4)
# Wrong results by construction; tools/diag/stepx_trace.py only reads the timestamps.
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_stepx_)
for variant in nostores; do
  case $variant in nostores) EXTRA="-DSTG_ABLATE_STORES";; esac
  FLAGS="-O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DSTG_STEPX_TRACE $EXTRA"
  mkdir -p ../../build/tracex_$variant
  for f in tgcn_stepx_fwd tgcn_stepx_bwd; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../build/tracex_$variant/$f.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/tracex_$variant/libstgraph_hip.so $OBJS ../../build/tracex_$variant/tgcn_stepx_fwd.o ../../build/tracex_$variant/tgcn_stepx_bwd.o -lhiprtc
done
ls -la ../../build/tracex_*/libstgraph_hip.so
