cd $GRAFT_REPO_ROOT
FPT_FUZZ_SEEDS=400 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz" 2>&1 | tail -4
