#!/bin/bash
# diagnostic (ablation build): the empirical-FDR pass with parts switched off
# 512: no Philox (made-up uniforms); 1024: no inverse-CDF draw; 4096: no ranking / histogram
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 512 1024 1536 4096 5632; do
  echo -n "ablate=$bits: "; FPT_ABLATE=$bits python3 tools/bench_fdr.py 2>&1 | tail -1
done
