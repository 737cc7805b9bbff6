#!/bin/bash
# Second library with per-wave phase timestamps in the folded TGCN forward step kernel (-DSTG_STEPX_TRACE) ->
# stgraph_amd/lib/diag/stepf_trace.so (travels to the GPU box; tools/diag/stepf_trace.py loads it through STGRAPH_AMD_LIB).
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
mkdir -p ../lib/diag ../../build/stepf_trace
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DSTG_STEPX_TRACE "$@" -c tgcn_stepf_fwd.hip -o ../../build/stepf_trace/tgcn_stepf_fwd.o
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_stepf_fwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/diag/stepf_trace.so $OBJS ../../build/stepf_trace/tgcn_stepf_fwd.o -lhiprtc
ls -la ../lib/diag/stepf_trace.so
