cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_q; mkdir -p $O
for mode in "FPT_SCAN_LEAN=0" "FPT_TABLE_LDS=1" "FPT_LEAN_TAB=0" "FPT_LEAN_PRIO=0"; do
  ( time env $mode python3 -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -6 ) > $O/$mode.log 2>&1
  echo "== $mode"; cat $O/$mode.log
done
