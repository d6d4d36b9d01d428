#!/bin/bash
# Same-box A/B of two ROUNDS: .ab_r02/ (git archive of round 2's final commit 22761b5, built in
# the build container; untracked) against the working tree, config 3, alternating, one lease.
#   mkdir .ab_r02 && git archive 22761b5 | tar -x -C .ab_r02 && make -C .ab_r02/footprint_tools_amd/csrc -j6 \
#     && make -C .ab_r02/oracle all          (then: gpurun -- 'bash tools/ab_r02.sh'; remove .ab_r02 afterwards)
# Attributes a round-over-round move of the driver's headline to the box or to the kernel.
cd "${GRAFT_REPO_ROOT:-.}"
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1 kernel_ms %.4f step_ms %.4f frac %.4f box_stream_GBps %s frac_of_box %s clock %s' % (r['kernel_ms'], d['ms_per_step'], r['frac'], r.get('box_stream_GBps'), r.get('frac_of_box'), r.get('shader_clock_GHz')))"; }
for rep in 1 2 3; do
  (cd .ab_r02 && python3 bench.py --config 3 --no-heavy --no-cpu-baseline --no-other-mode 2>/dev/null) | line "r02 "
  python3 bench.py --config 3 --no-heavy --no-cpu-baseline --no-other-mode --no-traffic-probe 2>/dev/null | line "HEAD"
done
