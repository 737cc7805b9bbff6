#!/bin/bash
# Second library with per-wave interval timestamps in the matrix-core TGCN step kernels (-DSTG_STEPX_TRACE):
# build/tracex/libstgraph_hip.so.  Run in the build container after `make -C stgraph_amd/csrc`; tools/diag/stepx_trace.py uses it.
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DSTG_STEPX_TRACE"
mkdir -p ../../build/tracex
for f in tgcn_stepx_fwd tgcn_stepx_bwd; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../build/tracex/$f.o; done
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_stepx_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/tracex/libstgraph_hip.so $OBJS ../../build/tracex/tgcn_stepx_fwd.o ../../build/tracex/tgcn_stepx_bwd.o -lhiprtc
ls -la ../../build/tracex/libstgraph_hip.so
