cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_m; mkdir -p $O
FPT_LEAN_BPL2=1 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan or fused or config1 or wave or tie or sparse" 2>&1 | tail -6 > $O/pytest_bpl2.log
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan or fused or config1 or wave or tie or sparse" 2>&1 | tail -3 > $O/pytest_bpl1.log
run() { # label cfg
  python3 bench.py --config $2 --no-heavy --no-cpu-baseline --no-traffic-probe --no-other-mode --no-config-legs --no-issue-probe --no-posterior --no-box-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 cfg$2 kernel_ms', round(d['roofline'].get('scan', d['roofline'])['kernel_ms'],4), 'step_ms', round(d['ms_per_step'],4), d.get('parity'))"
}
for rep in 1 2 3; do
  for cfg in 3 4; do
    for v in 0 1; do FPT_LEAN_BPL2=$v run bpl2=$v $cfg >> $O/ab.log 2>&1; done
  done
done
cat $O/pytest_bpl2.log $O/pytest_bpl1.log $O/ab.log
