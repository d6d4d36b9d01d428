cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_m; mkdir -p $O
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/bench_default.time; tail -c 300 $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_m/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_of_box'], r.get('valu_busy'), r.get('lds_busy'), r['traffic_over_algorithmic'])
for k,v in (d.get('configs') or {}).items():
    if v is None or 'error' in v: print(k, v); continue
    rr=v['roofline']; print(k, v['workload'], round(v['ms_per_step'],4), '%.4g'%v['value'], rr['bound'], rr['frac'], rr.get('valu_busy'), v.get('leg_wall_s'))
c5=d['configs']['5']; print({k:c5['roofline'].get(k) for k in ('frac','valu_busy','lds_busy','ta_busy','valu_instructions_per_draw','kernel_ms','fdr_pass_ms')})
p=c5['posterior']; print({k:p[k] for k in ('ms_per_launch_hip_events','dataset_bases_per_s','parity_max_abs_err','parity_ok')})
h=d['host_arrays']; print({k:(h[k]['value'], h[k]['ms_per_call']) for k in ('pageable','pinned')}, h['parity'])
print({k:v for k,v in d['cpu_baseline'].items() if not isinstance(v,(dict,str))})
PY
