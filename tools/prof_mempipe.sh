#!/bin/bash
# Where the memory pipeline of a kernel is busy or stalled: TA / TCP / TCC / SQ counters of bench.py, per kernel,
# per dispatch (mean).  Usage: bash tools/prof_mempipe.sh <tag> <kernel-substring> [bench args]
set -u
TAG=${1:-x}; KSUB=${2:-scan_lean}; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/mempipe_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-posterior --no-box-stream $@"
i=0
# (the TA stall and flat-wavefront counters are not collectable on this pool: those passes were dropped)
# (few counters of one block per pass: a request beyond what the hardware collects at once makes rocprofv3 abort
# and then hang in its signal handler -- every pass runs under its own timeout)
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_BUSY_avr TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($grp): failed or timed out"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    seen = set()
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "$KSUB" not in kn:
            continue
        short = kn.split("(")[0][-40:]
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = (f, row["Dispatch_Id"])
        if key not in seen and "End_Timestamp" in row:
            seen.add(key)
            dur[short].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for kn in sorted(agg):
    print(kn, "dispatch ns (under the profiler) mean %.0f" % (sum(dur[kn]) / max(1, len(dur[kn]))))
    for k in sorted(agg[kn]):
        v = agg[kn][k]
        print("   %-40s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
PY
