cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_b; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior"
for rep in 1 2; do
for p in 0 1 3 4 8 5 12 13 15; do
  for c in 3 4 2; do
    st=20; wu=5; [ $c != 3 ] && st=100 && wu=150
    FPT_LEAN_PRIO=$p python3 bench.py --config $c --steps $st --warmup $wu $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=$p cfg=$c ms=%.4f kernel_ms=%.4f parity=%s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']))" >> $O/prio.log
  done
done
done
cat $O/prio.log
