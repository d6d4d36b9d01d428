cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_za; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
for b in 1 3; do
  for c in 2 4; do
    FPT_LEAN_PRIO=$b python3 bench.py --config $c --steps 100 --warmup 150 $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blk=$b cfg=$c ms=%.4f kernel_ms=%.4f %s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['exp_bit_exact']))" >> $O/ab.log
  done
done
done
cat $O/ab.log
