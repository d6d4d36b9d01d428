cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_soak; mkdir -p $O
for i in 1 2 3; do
  python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/run$i.log; tail -1 $O/run$i.log
done
