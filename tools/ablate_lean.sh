#!/bin/bash
# diagnostic: time of the lean scan kernel stopped after phase A (1), B (2), C/D (3), or with
# phase E's normal cdf replaced by the identity (4); 0 = the whole kernel.  Ablation build only.
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in ${ABL_BITS:-1 2 3 4 0}; do
  FPT_ABLATE=$bits python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic-probe --no-other-mode "$@" 2>&1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stop=$bits kernel_ms=%.3f ms_per_step=%.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
done
