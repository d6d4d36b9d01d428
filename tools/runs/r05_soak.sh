cd $GRAFT_REPO_ROOT
FPT_FUZZ_SEEDS=150 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz" 2>&1 | tail -6
FPT_LEAN_BPL2=0 FPT_FUZZ_SEEDS=40 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean_kernel_fuzz or short_interval_fuzz or fused_scan_fuzz" 2>&1 | tail -3
