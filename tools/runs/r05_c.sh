cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c; mkdir -p $O
python3 -m pytest tests/test_ab_two_ranks.py -m gpu -x -q -k "many_ranks" 2>&1 | tail -80 > $O/pytest_ranks.log
python3 -m pytest tests -m gpu -x -q --deselect tests/test_ab_two_ranks.py::test_many_ranks_one_gpu_collectives 2>&1 | tail -30 > $O/pytest.log
AB_CFGS="3" bash tools/ab_lib.sh > $O/ab_defer.log 2>&1
cat $O/pytest_ranks.log | tail -70; cat $O/pytest.log | tail -15; cat $O/ab_defer.log
