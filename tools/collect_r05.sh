#!/bin/bash
# Round-5 evidence, collected on the GPU box in ONE call (outputs under gpurun_out/, copied into profiles/ afterwards):
#   rocprofv3 --kernel-trace --stats of the bench (whole, cfg4 alone, cfg2 alone, cfg3 alone), one --pmc pass (matrix-pipe counters) over
#   the folded TGCN step launches, the shared-tile A/B and phase trace, the split-form contraction microbench, the wave -> SIMD map.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
set -x
bash tools/prof_stats.sh r05full bench.py --no-live-pmc --no-cpu-baseline > gpurun_out/r05_prof_full.txt 2>&1
bash tools/prof_stats.sh r05tg bench.py --no-cora --no-gat --no-dynamic --no-cpu-baseline --no-live-pmc --tgcn-epochs 4 --steps 3 > gpurun_out/r05_prof_tg.txt 2>&1
bash tools/prof_stats.sh r05cfg2 bench.py --no-tgcn --no-cora --no-dynamic --no-gat --no-cpu-baseline --no-live-pmc --steps 30 > gpurun_out/r05_prof_cfg2.txt 2>&1
bash tools/prof_stats.sh r05gat tools/diag/gat_only.py > gpurun_out/r05_prof_gat.txt 2>&1
bash tools/prof_stats.sh r05dyn tools/diag/dyn_only.py rebuild_per_snapshot 40 > gpurun_out/r05_prof_dyn.txt 2>&1
# matrix-pipe counters of the folded step launches (program directly after "--": tools/pmc_mfma.sh)
STEP_VARIANTS=0 STEP_ITERS=10 STEP_WARM=2 bash tools/pmc_mfma.sh r05step tgcn_step tools/diag/step_coop_ab.py > gpurun_out/r05_pmc_step.txt 2>&1
# timing A/B (no profiler) + phase trace
STEP_VARIANTS=1,0,1,0 timeout -k 10 300 python tools/diag/step_coop_ab.py 2>/dev/null | tail -1 > gpurun_out/r05_step_coop_ab.json
STEP_VARIANTS=1,0,1,0 timeout -k 10 300 python tools/diag/step_coop_ab.py 25000 2>/dev/null | tail -1 >> gpurun_out/r05_step_coop_ab.json
for c in 1 0; do STGRAPH_AMD_LIB=$PWD/stgraph_amd/lib/diag/step_trace.so timeout -k 10 120 python tools/diag/step_trace2.py 50000 $c 2>/dev/null | tail -1; done > gpurun_out/r05_step_coop_trace.jsonl
timeout -k 10 400 python tools/microbench_gemm_x3.py 2>/dev/null | tail -1 > gpurun_out/r05_gemm_x3_microbench.json
stgraph_amd/lib/diag/wave_simd 12 > gpurun_out/r05_wave_simd.txt 2>&1
timeout -k 10 300 python tools/microbench_step.py 2>/dev/null | tail -1 > gpurun_out/r05_step_microbench.json
ls -la gpurun_out | tail -30
