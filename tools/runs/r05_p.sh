cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_p; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan or fused or config1 or short or tie or sparse or lean or predict" 2>&1 | tail -4 > $O/pytest.log
P=$PWD/footprint_tools_amd
AB_CFGS="4 2 3" AB_LIBS="A:$P/libfpt_hip_a.so main:$P/libfpt_hip.so" bash tools/ab_libs.sh > $O/ab_pp.log 2>&1
cat $O/pytest.log $O/ab_pp.log
