#!/bin/bash
# rocprofv3 kernel stats + SQ / TCC counters of the empirical-FDR pass (tools/bench_fdr.py:
# 20,000 x 500 bp, 100 null draws per base).  Usage: bash tools/prof_fdr.sh <tag>
set -u
TAG=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_fdr_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_fdr.py > $OUT/trace.log 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- python3 tools/bench_fdr.py > $OUT/pmc_$i.log 2>&1
done
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections
print(open("$OUT/trace.log").read().strip().splitlines()[-1])
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    print(open(f).read())
per = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "fdr_null" in row["Kernel_Name"]:
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("k_fdr_null, per dispatch (mean):")
for k in sorted(per):
    print("   %-24s n=%d mean=%.6g" % (k, len(per[k]), sum(per[k]) / len(per[k])))
PY
cat $OUT/summary.txt
