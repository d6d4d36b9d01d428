cd $GRAFT_REPO_ROOT
timeout 900 bash tools/prof_mempipe.sh r05_cfg3 "k_scan_lean" --config 3 --intervals 200000 > gpurun_out/r05_mempipe_cfg3.txt 2>&1
cat gpurun_out/r05_mempipe_cfg3.txt
