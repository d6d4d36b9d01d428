#!/bin/bash
# diagnostic (ablation build): vector / scalar instruction counts of k_fdr_null on the ragged shape
# (100,000 intervals, 100 draws per base) with parts switched off -- what each part issues.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in ${ABL_BITS:-0 512 1024 4096 5632}; do
  OUT=gpurun_out/pmc_fdrabl_$bits; mkdir -p $OUT
  FPT_ABLATE=$bits rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 ${ABL_SCRIPT:-tools/bench_fdr_ragged.py 100000 ${ABL_TIMES:-100}} > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = 0
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "fdr_null" in row["Kernel_Name"] or "fdr_slice" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
calls = 4.0  # one warm-up + three timed
print("ablate=$bits per call:", " ".join("%s=%.4g" % (k, v / calls) for k, v in sorted(tot.items())), open("$OUT/log.txt").read().strip().splitlines()[-1][-60:])
PY
done
