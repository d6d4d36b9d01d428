cd $GRAFT_REPO_ROOT
( time python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" 2>&1 | tail -15 ) 2>&1 | cut -c1-250
