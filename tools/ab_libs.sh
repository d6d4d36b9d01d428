#!/bin/bash
# A/B/... of several builds of the library on the same box, alternating: AB_LIBS="a:path b:path" AB_CFGS="3 2" bash tools/ab_libs.sh
# (label "main" = the tree's libfpt_hip.so)
run() { # label cfg lib
  local label=$1 cfg=$2 lib=$3
  FPT_LIB_PATH=$lib python3 bench.py --config $cfg --no-heavy --no-cpu-baseline --no-traffic-probe --no-other-mode --no-config-legs --no-issue-probe --no-posterior --no-box-stream ${AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label cfg$cfg kernel_ms', round(d['roofline'].get('scan', d['roofline'])['kernel_ms'],4), 'step_ms', round(d['ms_per_step'],4), d.get('parity'))"
}
for rep in 1 2 3; do
  for cfg in ${AB_CFGS:-3}; do
    for spec in ${AB_LIBS:-main:$PWD/footprint_tools_amd/libfpt_hip.so}; do
      run ${spec%%:*} $cfg ${spec#*:}
    done
  done
done
