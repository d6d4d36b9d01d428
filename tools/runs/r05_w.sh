cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_w; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "posterior or post" 2>&1 | tail -4 > $O/pytest.log
for i in 1 2 3; do python3 tools/bench_posterior.py 2>&1 | tail -1 | cut -c1-120 >> $O/post.log; done
python3 bench.py --config 5 --no-cpu-baseline --no-issue-probe --no-traffic-probe --no-box-stream --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['posterior']; print('cfg5 posterior', p['ms_per_launch_hip_events'], p['dataset_bases_per_s'], p['parity_max_abs_err'], p['parity_ok'])" >> $O/post.log 2>&1
cat $O/pytest.log $O/post.log
