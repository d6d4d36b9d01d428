cd $GRAFT_REPO_ROOT
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 1024 512 4096 1536 5632; do
  echo -n "ablate=$bits: "; FPT_ABLATE=$bits python3 tests/bench_fdr_ragged.py 2>&1 | tail -1
done
