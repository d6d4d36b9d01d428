cd $GRAFT_REPO_ROOT
export FPT_RCCL_LIB=$PWD/tests/fakerccl/libfakerccl.so FPT_COMM_TIMEOUT_S=120 FPT_LAUNCH_TIMEOUT_S=600
for args in "--config 3 --intervals 20000" "--config 5 --intervals 8000" "--config 2 --intervals 20000 --assembly gather" "--config 4 --intervals 20000 --gpus 4"; do
  g=2; case "$args" in *"--gpus 4"*) g=4; args="${args/ --gpus 4/}";; esac
  python3 bench.py --gpus $g $args --steps 3 --warmup 1 --share-gpu 2>/tmp/err.txt | python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); mg=d['multi_gpu']
print('$args gpus=$g -> n_gpus', d['n_gpus'], 'rccl_ranks', mg['rccl_ranks'], 'value %.3g'%d['value'], 'scan_only %.3g'%mg['scan_only']['value'], 'expected_vs_linear %.3f'%mg['expected_value_vs_linear'], d['parity'])" || tail -5 /tmp/err.txt
done
