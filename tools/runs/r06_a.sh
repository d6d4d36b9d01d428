cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a; mkdir -p $O
python3 -m pytest tests/test_ab_two_ranks.py -m gpu -x -q -k "launches" 2>&1 | tail -15 > $O/launch.log
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/launch.log $O/pytest.log $O/bench_default.time
tail -c 1500 $O/bench_default.err
