cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_i; mkdir -p $O
for i in 1 2 3; do python3 -m pytest tests/test_aa_bench_e2e.py tests/test_ab_two_ranks.py -m gpu -x -q 2>&1 | tail -3 >> $O/pytest_ranks.log; done
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/pytest_ranks.log; cat $O/pytest.log; cat $O/bench_default.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_i/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_box'])
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], v['roofline']['bound'], round(v['roofline']['frac'],4), v.get('leg_wall_s'), v['parity'])
p=d['configs']['5']['posterior']; print({k:p[k] for k in ('ms_per_launch_hip_events','dataset_bases_per_s','parity_max_abs_err','parity_ok')})
PY
