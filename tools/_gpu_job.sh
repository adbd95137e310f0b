#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/r06_gputests.log; tail -4 $O/r06_gputests.log
bash tools/refresh_profiles.sh r06 2>&1 | tail -3
