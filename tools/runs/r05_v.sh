cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_v; mkdir -p $O
for b in main abl1 abl2 abl4 abl8 abl16 abl31; do
  L=$PWD/footprint_tools_amd/libfpt_hip.so; [ $b != main ] && L=$PWD/footprint_tools_amd/libfpt_hip_$b.so
  echo -n "$b: " >> $O/abl.log
  FPT_LIB_PATH=$L python3 tools/bench_posterior.py 2>&1 | tail -1 | cut -c1-110 >> $O/abl.log
  FPT_LIB_PATH=$L timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p_$b -- python3 tools/bench_posterior.py > $O/p_$b.log 2>&1
  python3 - "$O/p_$b" >> $O/abl.log <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_posterior<" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
if m: print("    VALU insts %.4g  active %.4g  lanes %.3f" % (m["SQ_INSTS_VALU"], m["SQ_ACTIVE_INST_VALU"], m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"])))
PY
done
cat $O/abl.log
