cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior"
for rep in 1 2; do
for p in 0 3000 3300 3200 3100 3210 3110 3321 3330 3333 2000 1000 2100 3003 3303 123 3030 3313; do
  for c in 3 4 2; do
    st=20; wu=5; [ $c != 3 ] && st=100 && wu=150
    FPT_LEAN_PRIO=$p python3 bench.py --config $c --steps $st --warmup $wu $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=$p cfg=$c ms=%.4f kernel_ms=%.4f ok=%s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['exp_bit_exact']))" >> $O/prio.log
  done
done
done
python3 - <<'PY'
import re,collections
r=collections.defaultdict(list)
for ln in open('gpurun_out/r06_c/prio.log'):
    m=re.match(r'prio=(\d+) cfg=(\d) ms=([\d.]+) kernel_ms=([\d.]+)',ln)
    r[(int(m.group(1)),m.group(2))].append(float(m.group(4)))
ps=sorted(set(k[0] for k in r))
for p in ps: print('%04d'%p, ' '.join('cfg%s %s'%(c, '/'.join('%.4f'%x for x in r[(p,c)])) for c in '342'))
PY
