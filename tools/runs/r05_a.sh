cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_a; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 512 1024 2048 4096 5120 7680; do
  echo -n "slices=0 ablate=$bits: " >> $O/ablate.log; FPT_FDR_SLICES=0 FPT_ABLATE=$bits python3 tools/bench_fdr_ragged.py 100000 100 2>&1 | tail -1 >> $O/ablate.log
done
unset FPT_LIB_PATH
cat $O/pytest.log; cat $O/bench_default.time; tail -c 1500 $O/bench_default.err; cat $O/ablate.log
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_a/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'])
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    print(k, v['workload'], v['ms_per_step'], v['value'], v['roofline']['bound'], v['roofline']['frac'], v.get('leg_wall_s'), v['parity'])
print(json.dumps(d['configs']['5']['roofline'], indent=1)[:2500])
print(json.dumps(d['cpu_baseline'], indent=1)[:2500])
PY
