cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_n; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.log
FPT_LEAN_BPL2=0 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan or fused or config1 or short or tie or sparse or lean" 2>&1 | tail -3 > $O/pytest_bpl1.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/pytest.log $O/pytest_bpl1.log $O/bench_default.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_n/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_box'], d['roofline']['kernel'])
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], v['roofline']['bound'], round(v['roofline']['frac'],4), v.get('leg_wall_s'), v['parity'])
PY
