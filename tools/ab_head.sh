#!/bin/bash
# A/B on the same box: libfpt_hip_a.so (git HEAD, tools/ab_build.sh) vs libfpt_hip.so (working tree)
run() { # label cfg env...
  local label=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --no-heavy --no-cpu-baseline --no-traffic-probe --no-other-mode --no-config-legs --no-issue-probe --no-posterior --no-box-stream ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label cfg$cfg kernel_ms', round(d['roofline']['kernel_ms'],4), 'step_ms', round(d['ms_per_step'],4), d.get('parity'))"
}
for rep in 1 2 3; do
  for cfg in ${AB_CFGS:-3}; do
    run A $cfg FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_a.so
    run B $cfg FPT_X=0
  done
done
