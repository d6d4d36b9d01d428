#!/bin/bash
# VALU / SALU / LDS instruction counts of the fused scan kernel per phase: the ablate build
# returns after phase A (32), B (64), C (128) or skips E (256); differences give the phases.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
OUT=gpurun_out/phase
mkdir -p $OUT
for bits in 32 64 128 256 0; do
  FPT_ABLATE=$bits rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/b$bits -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --nb-mode memo "$@" > $OUT/b$bits.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for bits in (32, 64, 128, 256, 0):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/b%d/*/*_counter_collection.csv" % bits):
        for row in csv.DictReader(open(f)):
            if "scan_fused" in row["Kernel_Name"] and "true, true>" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    w = sum(agg["SQ_WAVES"]) / max(len(agg["SQ_WAVES"]), 1)
    print("ablate=%-4d per wave:" % bits, "  ".join("%s=%.1f" % (k[3:], (sum(v) / len(v)) / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
