#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "aside or fused_softmax or split_update or train_step or bit" 2>&1 | tail -5
for v in 1 0 1; do echo "tail_aside=$v"; FSMG_TAIL_ASIDE=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o st -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs > /dev/null 2>&1; python $R/tools/step_timeline.py $(find /tmp/prof_f -name "*.db" | head -1) 40 > $O/r05q_timeline.txt 2>&1
cat $O/r05q_timeline.txt
