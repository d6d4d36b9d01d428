cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_x; mkdir -p $O
for rep in 1 2; do for b in main NO_LOOP4 NO_FAST_FITS; do
  L=$PWD/footprint_tools_amd/libfpt_hip.so; [ $b != main ] && L=$PWD/footprint_tools_amd/libfpt_hip_$b.so
  echo -n "$b: " >> $O/ab.log; FPT_LIB_PATH=$L python3 tools/bench_posterior.py 2>&1 | tail -1 | cut -c1-110 >> $O/ab.log
  FPT_LIB_PATH=$L python3 bench.py --config 5 --no-cpu-baseline --no-issue-probe --no-traffic-probe --no-box-stream --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['posterior']; print('   $b cfg5 posterior', round(p['ms_per_launch_hip_events'],3))" >> $O/ab.log 2>&1
done; done
cat $O/ab.log
