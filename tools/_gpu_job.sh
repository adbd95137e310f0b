#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused_softmax or eval or validate" 2>&1 | tail -5
for v in 0 1 0 1; do echo "fused=$v"; FSMG_FUSED_SOFTMAX=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
for v in 1; do rm -rf /tmp/prof_f$v; FSMG_FUSED_SOFTMAX=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f$v -o st -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs > /dev/null 2>&1; python $R/tools/step_timeline.py $(find /tmp/prof_f$v -name "*.db" | head -1) 40 > $O/r05o_fused${v}_timeline.txt 2>&1; done
cat $O/r05o_fused1_timeline.txt
