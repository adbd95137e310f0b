cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r03 > gpurun_out/r03_refresh.log 2>&1
tail -5 gpurun_out/r03_refresh.log; ls -la gpurun_out | grep r03_ | grep -v probe | tail -30
