#!/bin/bash
# quick PMC profile of bench.py (SQ counters only).  Usage: bash tools/prof_pmc.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-config-legs --no-issue-probe "$@" > $OUT/$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "scan_fused" in kn or "scan_lean" in kn or "fdr_null" in kn:
            m = re.search(r"<(.*?)>", kn)
            agg[kn.split("<")[0].split("(")[0] + "<" + (m.group(1) if m else "") + ">"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kn in sorted(agg):
    print(kn)
    for k in sorted(agg[kn]):
        v = agg[kn][k]
        print("   %-24s n=%d mean=%.4g" % (k, len(v), sum(v)/len(v)))
PY
