#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes for bench.py.
# Usage: bash tests/prof_run.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 5 --warmup 1 --no-cpu-baseline}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ${PROF_EXTRA:-} > $OUT/pmc_$name.log 2>&1
  echo "pmc $grp rc=$?" >> $OUT/trace.log
done
find $OUT -name "*.csv" | head -50 >> $OUT/trace.log
