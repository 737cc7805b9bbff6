#!/bin/bash
# The folded forward step kernel with 8 / 12 / 16 waves per workgroup -> stgraph_amd/lib/diag/stepf_w<W>.so (timed on the GPU box with
# STGRAPH_AMD_LIB=... python tools/microbench_step.py).
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
mkdir -p ../lib/diag ../../build/stepf_var
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_stepf_fwd)
EXTRA=${STEPF_EXTRA:-}
for w in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DSTG_STEPF_WAVES=$w $EXTRA -c tgcn_stepf_fwd.hip -o ../../build/stepf_var/w$w.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/diag/stepf_w$w.so $OBJS ../../build/stepf_var/w$w.o -lhiprtc
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DSTG_STEPF_WAVES=$w $EXTRA -S --cuda-device-only -o /tmp/sf_w$w.s tgcn_stepf_fwd.hip 2>/dev/null
  echo "waves $w: $(grep -m2 'vgpr_count\|vgpr_spill' /tmp/sf_w$w.s | tr -d '\n')"
done
