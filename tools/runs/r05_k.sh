cd $GRAFT_REPO_ROOT
for lib in libfpt_hip.so libfpt_hip_b.so libfpt_hip_c.so; do
 echo "== $lib"; FPT_LIB_PATH=$PWD/footprint_tools_amd/$lib python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "posterior_driver_against" 2>&1 | grep "AssertionError:\|passed\|failed" | tail -3
done
echo "== main, tables off"; FPT_POSTERIOR_TABLES=0 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "posterior_driver_against" 2>&1 | grep "AssertionError:\|passed\|failed" | tail -3
