#!/bin/bash
# diagnostic: VALU / LDS instruction counts per cumulative phase (ablation build)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in ${ABL_BITS:-32 64 128 256 0}; do
  rm -rf /tmp/abl_pmc
  FPT_ABLATE=$bits rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/abl_pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --nb-mode memo > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/abl_pmc/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "scan_fused" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("ablate=$bits", " ".join("%s=%.0f" % (k.replace("SQ_",""), sum(v)/len(v)/8e5) for k, v in sorted(agg.items())), "(per wavefront)")
PY
done
