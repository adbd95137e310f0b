#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
V="h1: h2: h3: h4: h5: h6: h7: h8: h9:"
echo "== default GPU_MAX_HW_QUEUES, with the probe"; timeout 300 python tools/ab_step.py --rounds 3 --steps 20 $V 2>/dev/null | grep "episodes/s"
echo "== FSMG_AUX_TRIES=1 (no redraw)"; FSMG_AUX_TRIES=1 timeout 300 python tools/ab_step.py --rounds 3 --steps 20 $V 2>&1 | grep "episodes/s\|fsmg\]"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "second_stream or all_schedules or xcd_partitioned_schedule_gives" -p no:cacheprovider 2>&1 | tail -5
