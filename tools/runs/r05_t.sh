cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or full_size or short or fuzz" 2>&1 | tail -3 > $O/pytest.log
run() { python3 bench.py --config $2 --no-heavy --no-cpu-baseline --no-traffic-probe --no-other-mode --no-config-legs --no-issue-probe --no-posterior --no-box-stream --warmup 100 --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline'].get('scan', d['roofline']); print('$1 cfg$2 kernel_ms', round(r['kernel_ms'],4), 'seq_ms', round(r['launch_sequence_ms'],4), 'step_ms', round(d['ms_per_step'],4), d['parity']['exp_bit_exact'])"; }
for rep in 1 2 3; do for v in 0 1; do FPT_SCAN_STREAMS=$v run streams=$v 4 >> $O/ab.log 2>&1; done; done
cat $O/pytest.log $O/ab.log
