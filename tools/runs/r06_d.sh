cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_d; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "posterior" 2>&1 | tail -8 > $O/pytest_post.log
cat $O/pytest_post.log
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream"
for rep in 1 2 3; do
for p in 0 1; do
  FPT_POST_PRIO=$p python3 bench.py --config 5 --steps 4 --warmup 2 $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['posterior']; print('postprio=$p step_ms=%.3f post_ms=%.4f dsb/s=%.4g parity=%s %s'%(d['ms_per_step'], p['ms_per_launch_hip_events'], p['dataset_bases_per_s'], p['parity_ok'], p['parity_max_abs_err']))" >> $O/post.log
done
done
cat $O/post.log
