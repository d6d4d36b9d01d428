cd $GRAFT_REPO_ROOT
bash tools/prof_post.sh r05a > gpurun_out/r05_e_post.log 2>&1
tail -40 gpurun_out/r05_e_post.log
