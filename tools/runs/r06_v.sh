cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_v; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; cat $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; cat $O/bench_default.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_v/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'))
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; print(k, round(v['ms_per_step'],4), '%.4g'%v['value'], round(rr['frac'],4))
h=d['host_arrays']; print({k:(h[k]['value'], h[k]['ms_per_call']) for k in ('pageable','pinned')})
PY
