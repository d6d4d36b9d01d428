#!/bin/bash
# diagnostic: average shader clock of the lean kernel = GRBM_GUI_ACTIVE / dispatch duration,
# with (FPT_ABLATE=0) and without (6) its memory traffic.  Ablation build.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for ab in 0 6 5; do
  OUT=gpurun_out/clock_$ab
  rm -rf $OUT; mkdir -p $OUT
  FPT_ABLATE=$ab rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "scan_lean" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    if rows:
        print("ablate=$ab columns:", [k for k in rows[0].keys()][:24])
        for r in rows[:4]:
            dur = (int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0))) if "End_Timestamp" in r else 0
            v = float(r["Counter_Value"])
            print("  GRBM_GUI_ACTIVE %.4g  duration %d ns  -> %.3f GHz (per-XCD sum / 8: %.3f)" % (v, dur, v / max(dur, 1), v / 8 / max(dur, 1)))
PY
done
