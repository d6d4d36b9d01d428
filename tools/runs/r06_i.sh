cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_i; mkdir -p $O
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
export FPT_FDR_SLICES=0
# 32768 cheap hash for Philox, 65536 no z stores, 131072 no barriers in a pass; 1024 no gathers, 2048 no window sums, 4096 no ranking
for rep in 1 2; do
for bits in 0 32768 65536 131072 98304 229376 7168 39936 236544; do
  FPT_ABLATE=$bits python3 tools/bench_fdr_ragged.py 100000 100 2>/dev/null | sed "s/^/slices=0 ablate=$bits: /" >> $O/ablate.log
done
done
cat $O/ablate.log
