#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in ${ABL_BITS:-0 32 64 128 256}; do
  FPT_ABLATE=$bits python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic-probe --nb-mode memo ${ABL_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ablate=$bits kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
done
