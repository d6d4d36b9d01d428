cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_x; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
for b in 1 2; do
    FPT_LEAN_PRIO=$b python3 bench.py --config 3 --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('multi=$b cfg=3 ms=%.4f kernel_ms=%.4f %s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']))" >> $O/ab.log
done
done
cat $O/ab.log
