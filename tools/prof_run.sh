#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):
#   bash tools/prof_run.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats of bench.py            -> gpurun_out/prof_<tag>/trace
# 2. separate --pmc passes (never combined with a trace):    -> gpurun_out/prof_<tag>/pmc_*
#    FETCH_SIZE | WRITE_SIZE | SQ issue/wait counters | LDS counters
# 3. a text summary                                           -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-r01}; shift || true
ARGS="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe $@"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 $ARGS > $OUT/trace.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/pmc_$i.log 2>&1
done
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections, json
print("command: python3 bench.py --steps 10 --warmup 2 $ARGS")
for line in open("$OUT/trace.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("bench: value=%.4g %s ms_per_step=%.4f kernel_ms=%.4f roofline.frac=%.4f" % (
            d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
print()
print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    print(open(f).read())
print("== rocprofv3 --pmc passes, per dispatch (mean over dispatches), by kernel")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "scan_fused" in row["Kernel_Name"] or "scan_lean" in row["Kernel_Name"]:
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
agg = collections.defaultdict(float)
for name in sorted(per):
    print("--", name)
    for k in sorted(per[name]):
        v = per[name][k]
        print("   %-24s n=%d mean=%.6g min=%.6g max=%.6g" % (k, len(v), sum(v)/len(v), min(v), max(v)))
        agg[k] += sum(v) / len(v)
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f, w = agg["FETCH_SIZE"], agg["WRITE_SIZE"]
    print()
    print("(summed over the scan kernels of one step: k_scan_lean + the redo instance of k_scan_fused)")
    print("HBM traffic per launch (MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB;")
    print("on gfx950 FETCH_SIZE reports half the bytes of a coalesced streaming read -> doubled):")
    print("  read  = 2 * %.6g KiB = %.4f GB" % (f, 2 * f * 1024 / 1e9))
    print("  write =     %.6g KiB = %.4f GB" % (w, w * 1024 / 1e9))
    print("  total = %.4f GB" % ((2 * f + w) * 1024 / 1e9))
PY
cat $OUT/summary.txt
