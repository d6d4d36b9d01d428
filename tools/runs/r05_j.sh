cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "posterior" 2>&1 | grep -v "^E  *+\|array(" | tail -40 | cut -c1-300
