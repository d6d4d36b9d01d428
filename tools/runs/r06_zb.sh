cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_zb; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
  python3 tools/bench_fdr_ragged.py 100000 100 2>/dev/null >> $O/fdr.log
  python3 bench.py --config 5 --steps 4 --warmup 2 $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 step_ms=%.3f fdr_ms=%.3f parity=%s'%(d['ms_per_step'], d['fdr']['ms_per_call'], d['parity']))" >> $O/fdr.log
done
cat $O/fdr.log
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fdr or full_size_config4" 2>&1 | tail -3
