#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x; FSMG_XCD_OVERLAP=1 FSMG_XOV_DW_SPLIT=${XS:-12} FSMG_XOV_BLOCKS=${XB:-2} rocprofv3 --kernel-trace -d /tmp/prof_x -o st -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-breakdown > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_x -name "*.db" | head -1) 25 > $O/xov_step_timeline.txt 2>&1
tail -60 $O/xov_step_timeline.txt
