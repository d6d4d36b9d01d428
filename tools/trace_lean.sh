#!/bin/bash
# diagnostic (ablation build): per-workgroup timeline of the lean kernel on config 3 and 2
cd "${GRAFT_REPO_ROOT:-.}"
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for cfg in ${TRACE_CFGS:-3 2 4}; do
  for ab in 0 6; do
    rm -f /tmp/trace.bin.*
    FPT_ABLATE=$ab FPT_LEAN_TRACE=/tmp/trace.bin python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy > /dev/null 2>&1
    for f in /tmp/trace.bin.*; do
      nt=${f##*.}
      echo "== config $cfg ablate=$ab workgroup size $nt"
      python3 tools/lean_trace.py $f $nt
    done
  done
done
