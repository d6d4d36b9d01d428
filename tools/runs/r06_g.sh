cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_g; mkdir -p $O
python3 tools/bench_host_arrays.py 0,4194304,8388608,16777216,1048576 > $O/host.log 2>&1
cat $O/host.log
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
cat $O/pytest.log
