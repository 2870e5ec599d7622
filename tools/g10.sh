cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof_r03.sh > gpurun_out/prof_r03.log 2>&1
tail -30 gpurun_out/prof_r03.log
