cd $GRAFT_REPO_ROOT
timeout 300 bash tools/prof_post.sh r05 > gpurun_out/r05_prof_post.log 2>&1
timeout 600 bash tools/prof_run.sh r05_cfg5 --config 5 > gpurun_out/r05_prof_cfg5.log 2>&1
tail -22 gpurun_out/r05_prof_post.log
