cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
for c in 3 2 4 5; do timeout 600 bash tools/prof_run.sh r06_cfg$c --config $c --no-host-arrays > $O/prof_cfg$c.log 2>&1; done
timeout 300 bash tools/prof_post.sh r06 > $O/prof_post.log 2>&1
cat $O/pytest.log $O/bench_default.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'), r.get('lds_busy'))
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], rr['bound'], round(rr['frac'],4), rr.get('valu_busy'), v.get('leg_wall_s'))
p=d['configs']['5']['posterior']; print({k:p[k] for k in ('ms_per_launch_hip_events','dataset_bases_per_s','parity_max_abs_err','parity_ok')})
h=d['host_arrays']; print({k:(h[k]['value'], h[k]['ms_per_call']) for k in ('pageable','pinned')}, h['parity'])
PY
tail -3 $O/prof_post.log
