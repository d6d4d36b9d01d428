cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_d; mkdir -p $O
P=$PWD/footprint_tools_amd
AB_CFGS="3" AB_LIBS="main:$P/libfpt_hip.so bunch1:$P/libfpt_hip_b.so bunch2:$P/libfpt_hip_c.so" bash tools/ab_libs.sh > $O/ab_bunch.log 2>&1
python3 -m pytest tests -m gpu -x -q -k "posterior or log_fast or special" 2>&1 | tail -12 > $O/pytest.log
python3 tools/bench_posterior.py > $O/posterior.log 2>&1
python3 bench.py --config 5 --no-cpu-baseline --no-issue-probe --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['posterior']; print('cfg5 posterior', p['ms_per_launch_hip_events'], p['dataset_bases_per_s'], p['parity_max_abs_err'], p['parity_ok'])" > $O/post5.log 2>&1
cat $O/ab_bunch.log; cat $O/pytest.log; tail -2 $O/posterior.log; cat $O/post5.log
