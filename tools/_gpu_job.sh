cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r03 > gpurun_out/r03_refresh.log 2>&1
tail -3 gpurun_out/r03_refresh.log | cut -c1-400
