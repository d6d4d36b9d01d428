cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_p; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
for lib in libfpt_hip_A.so libfpt_hip.so; do
  FPT_LIB_PATH=$PWD/footprint_tools_amd/$lib python3 bench.py --config 3 --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg=3 ms=%.4f kernel_ms=%.4f parity=%s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']))" >> $O/ab.log
done
done
cat $O/ab.log
for lib in libfpt_hip_A.so libfpt_hip.so; do
FPT_LIB_PATH=$PWD/footprint_tools_amd/$lib timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc_$lib -- python3 bench.py --config 3 --intervals 200000 --steps 3 --warmup 1 $F > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
a=collections.defaultdict(list)
for f in glob.glob("$O/pmc_$lib/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'scan_lean' in r['Kernel_Name']: a[r['Counter_Name']].append(float(r['Counter_Value']))
print("$lib", {k: '%.4g'%(sum(v)/len(v)) for k,v in a.items()})
PY
done
