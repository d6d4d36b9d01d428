#!/bin/bash
# diagnostic (ablation build): vector / scalar / LDS instructions of k_scan_lean per wavefront with the
# kernel cut short (FPT_ABLATE = lean stop code: 1 after staging, 2 after phase B, 3 without the
# Stouffer windows, 4 windows without ndtr, 0 whole).  Config 3 by default.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for stop in ${ABL_STOPS:-1 2 3 4 0}; do
  OUT=gpurun_out/pmc_lean_$stop; rm -rf $OUT; mkdir -p $OUT
  FPT_ABLATE=$stop rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -- python3 bench.py --config ${ABL_CFG:-3} --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "scan_lean" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
w = tot.get("SQ_WAVES", 1.0)
print("stop=$stop per wavefront:", " ".join("%s=%.1f" % (k.replace("SQ_INSTS_", ""), v / w) for k, v in sorted(tot.items()) if k != "SQ_WAVES"), "waves=%.3g" % w)
PY
done
