cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_q; mkdir -p $O
B=$PWD/footprint_tools_amd/libfpt_hip_b.so
FPT_LIB_PATH=$B python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fdr" 2>&1 | tail -5 > $O/pytest_b.log
for rep in 1 2 3; do
  echo -n "main: " >> $O/ab.log; python3 tools/bench_fdr_ragged.py 100000 100 2>&1 | tail -1 >> $O/ab.log
  echo -n "wslice: " >> $O/ab.log; FPT_LIB_PATH=$B python3 tools/bench_fdr_ragged.py 100000 100 2>&1 | tail -1 >> $O/ab.log
done
for rep in 1 2; do for lib in main wslice; do
  L=$PWD/footprint_tools_amd/libfpt_hip.so; [ $lib = wslice ] && L=$B
  FPT_LIB_PATH=$L python3 bench.py --config 5 --no-cpu-baseline --no-issue-probe --no-traffic-probe --no-posterior --no-box-stream --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib cfg5 step_ms', round(d['ms_per_step'],3), 'fdr_ms', round(d['fdr']['ms_per_call'],3), d['parity'])" >> $O/ab.log
done; done
cat $O/pytest_b.log $O/ab.log
