#!/bin/bash
# Second library with per-wave phase timestamps in the TGCN step kernels (-DSTG_STEP_TRACE): stgraph_amd/lib/diag/step_trace.so (travels to the GPU box; load it with STGRAPH_AMD_LIB).
# Run in the build container after `make -C stgraph_amd/csrc`; tools/diag/step_trace.py uses it on the GPU box.
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
FLAGS="-O3 --offload-arch=gfx950 --offload-compress -fPIC -ffp-contract=off -std=c++17 -DSTG_STEP_TRACE"
mkdir -p ../../build/trace
for f in tgcn_step_fwd tgcn_step_bwd; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../build/trace/$f.o; done
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_step_)
mkdir -p ../lib/diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/diag/step_trace.so $OBJS ../../build/trace/tgcn_step_fwd.o ../../build/trace/tgcn_step_bwd.o -lhiprtc
ls -la ../lib/diag/step_trace.so
