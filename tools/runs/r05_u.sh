cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_u; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/pytest.log $O/smoke.log $O/bench_default.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_u/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'), r.get('traffic_over_algorithmic'))
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; sr=rr.get('scan', rr); print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], rr['bound'], round(rr['frac'],4), sr.get('valu_busy'), sr.get('traffic_over_algorithmic'))
PY
