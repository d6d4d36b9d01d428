#!/bin/bash
# A/B of the lean kernel's start stagger (FPT_LEAN_STAGGER_US): kernel ms of config 3 and 2
mkdir -p gpurun_out
run() { # label cfg env...
  local label=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --no-heavy --no-cpu-baseline --no-other-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label cfg$cfg', round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4))"
}
for rep in 1 2; do
  run off 3 FPT_X=0
  for us in 4 8 13 20; do run us$us 3 FPT_LEAN_STAGGER_US=$us; done
  run us13n1024 3 FPT_LEAN_STAGGER_US=13 FPT_LEAN_STAGGER_N=1024
  run off 2 FPT_X=0
  for us in 3 6 10 15; do run us$us 2 FPT_LEAN_STAGGER_US=$us; done
done
