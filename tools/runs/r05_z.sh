cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_z; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python3 examples/synthetic_detect.py > $O/example.log 2>&1; echo "example rc $?" >> $O/example.log
( time python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cat $O/pytest.log $O/smoke.log; tail -4 $O/example.log; cat $O/bench.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_z/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'), r.get('traffic_over_algorithmic'))
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; print(k, round(v['ms_per_step'],4), '%.4g'%v['value'], rr['bound'], round(rr['frac'],4))
p=d['configs']['5']['posterior']; print(p['ms_per_launch_hip_events'], p['dataset_bases_per_s'], p['parity_ok'])
PY
