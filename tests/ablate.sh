#!/bin/bash
# diagnostic: phase ablation of the fused kernel (timing only, results are wrong by design)
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for mode in memo direct; do
for bits in 0 1 2 4 8 16 3 7 15 31; do
  FPT_ABLATE=$bits python bench.py --steps 5 --warmup 1 --no-cpu-baseline --nb-mode $mode ${ABL_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('mode=$mode ablate=$bits kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
done; done
