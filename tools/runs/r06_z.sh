cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_z; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
  for c in 3 4; do
    st=100; wu=150; [ $c = 3 ] && st=20 && wu=5
    python3 bench.py --config $c --steps $st --warmup $wu $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg=$c ms=%.4f kernel_ms=%.4f %s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']))" >> $O/ab.log
  done
done
cat $O/ab.log
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
