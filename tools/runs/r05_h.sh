cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h; mkdir -p $O
python3 -m pytest tests/test_aa_bench_e2e.py tests/test_ab_two_ranks.py -m gpu -x -q 2>&1 | tail -150 > $O/pytest_ranks.log
P=$PWD/footprint_tools_amd
AB_CFGS="4 2 3" AB_LIBS="A:$P/libfpt_hip_a.so main:$P/libfpt_hip.so" bash tools/ab_libs.sh > $O/ab_gather.log 2>&1
cat $O/pytest_ranks.log | grep -v "^E   *File\|^E *$" | tail -120; cat $O/ab_gather.log
