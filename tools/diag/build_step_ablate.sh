#!/bin/bash
# Diagnosis libraries: the step kernels without their row-piece stores (build/ablate/nostore.so) and with one LDS weight fragment
# reused for every product step (build/ablate/nolds.so).  Results are wrong by construction; only the timings mean anything.
set -e
cd "$(dirname "$0")/../../stgraph_amd/csrc"
mkdir -p ../../build/ablate
OBJS=$(ls ../../build/obj/*.o | grep -v tgcn_step_)
for v in nostore:-DSTG_ABLATE_STORES nolds:-DSTG_ABLATE_LDS; do
  name=${v%%:*}; flag=${v##*:}
  for f in tgcn_step_fwd tgcn_step_bwd; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 $flag -c $f.hip -o ../../build/ablate/${f}_$name.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ablate/$name.so $OBJS ../../build/ablate/tgcn_step_fwd_$name.o ../../build/ablate/tgcn_step_bwd_$name.o -lhiprtc
done
ls -la ../../build/ablate/*.so
