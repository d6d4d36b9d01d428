cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_f; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan_host_pipeline" 2>&1 | tail -15 > $O/pytest_host.log
cat $O/pytest_host.log
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
cat $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/bench_default.time; tail -c 600 $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_f/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'), r.get('lds_busy'))
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], rr['bound'], rr['frac'], v.get('leg_wall_s'))
print(json.dumps(d['host_arrays'], indent=1))
print(d['other_nb_mode'], d['heavy_tailed']['value'], d['heavy_tailed']['cold']['value'], d['sparse_counts']['value'])
PY
