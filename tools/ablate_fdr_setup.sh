#!/bin/bash
# diagnostic (ablation build): the per-interval set-up of the FDR kernel (4 draws per base: one pass)
# with parts switched off.  8192: thresholds = ndtri(P), no search; 16384: no sort; 65536: observed
# windows not re-made (obs path); 1024+4096: no draw / no ranking in the one pass
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 8192 16384 24576 5120 29696; do
  echo -n "ablate=$bits: "; FPT_ABLATE=$bits python3 tools/bench_fdr_ragged.py 100000 4 2>&1 | tail -1
done
