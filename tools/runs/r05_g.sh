cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_g; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
for rep in 1 2; do for lib in libfpt_hip_a.so libfpt_hip.so; do
  echo -n "$lib: " >> $O/post_ab.log; FPT_LIB_PATH=$PWD/footprint_tools_amd/$lib python3 tools/bench_posterior.py 2>&1 | tail -1 >> $O/post_ab.log
done; done
python3 bench.py --config 5 --no-cpu-baseline --no-issue-probe --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['posterior']; print('cfg5 posterior', p['ms_per_launch_hip_events'], p['dataset_bases_per_s'], p['parity_max_abs_err'], p['parity_ok'])" > $O/post5.log 2>&1
cat $O/pytest.log; cat $O/post_ab.log; cat $O/post5.log
