#!/bin/bash
# diagnostic (ablation build): bench.py --config 5 (ragged whole-genome shape, scan + FDR) with parts of
# the FDR kernel switched off.  512: no Philox; 1024: no inverse-CDF draw; 4096: no ranking
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 512 1024 4096 5632; do
  echo -n "ablate=$bits: "; FPT_ABLATE=$bits python3 bench.py --config 5 --steps 3 --warmup 1 --no-posterior --no-cpu-baseline --no-traffic-probe --no-other-mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2), 'ms/step', '%.3g bases/s' % d['value'])"
done
