cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_k; mkdir -p $O
ARGS="--config 3 --intervals 50000 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-posterior --no-box-stream --no-host-arrays"
i=0
for prio in 1 0; do
for grp in "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum" \
           "TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  FPT_LEAN_PRIO=$prio timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 bench.py $ARGS > $O/p$i.log 2>&1 || echo "prio=$prio pass $i ($grp): failed or timed out" | tee -a $O/failed.txt
  echo "$i prio=$prio $grp" >> $O/passes.txt
done
done
python3 - <<'PY'
import csv, glob, collections, re
passes = {}
for ln in open("gpurun_out/r06_k/passes.txt"):
    a = ln.split()
    passes["p" + a[0]] = a[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r06_k/p*/*/*_counter_collection.csv"):
    pn = re.search(r"/(p\d+)/", f).group(1)
    seen = set()
    for row in csv.DictReader(open(f)):
        if "scan_lean" not in row["Kernel_Name"]:
            continue
        agg[passes[pn]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = (f, row["Dispatch_Id"])
        if key not in seen and "End_Timestamp" in row:
            seen.add(key)
            dur[passes[pn]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for pr in sorted(agg):
    print("k_scan_lean<512,2>, 50,000 x 1 kb x 5 scales, FPT_LEAN_%s: dispatch ns under the profiler, mean %.0f" % (pr, sum(dur[pr]) / max(1, len(dur[pr]))))
    for k in sorted(agg[pr]):
        v = agg[pr][k]
        print("   %-40s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
PY
cat $O/failed.txt 2>/dev/null
