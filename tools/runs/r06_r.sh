cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_r; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
for rep in 1 2 3; do
for lib in libfpt_hip.so libfpt_hip_pk.so; do
  for c in 3 4 2; do
    st=20; wu=5; [ $c != 3 ] && st=100 && wu=150
    FPT_LIB_PATH=$PWD/footprint_tools_amd/$lib python3 bench.py --config $c --steps $st --warmup $wu $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg=$c ms=%.4f kernel_ms=%.4f parity=%s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']))" >> $O/ab.log
  done
done
done
cat $O/ab.log
FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_pk.so python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
