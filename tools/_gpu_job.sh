#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
