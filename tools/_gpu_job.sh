#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05g}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "half_row or xcd_partitioned_schedule_gives or partitioned_schedule_times_out" -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
cd /tmp && export TMPDIR=/tmp
run() {  # name env...
  n=$1; shift
  for rep in 1 2; do env "$@" python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d['guard']['timeouts'])" >> $O/${TAG}_values.txt; done
}
run base FSMG_NOP=1
run half2 FSMG_XOV_HALF=2
run half3 FSMG_XOV_HALF=3
run half5 FSMG_XOV_HALF=5
run half8 FSMG_XOV_HALF=8
run half23 FSMG_XOV_HALF=23
run base FSMG_NOP=1
rm -rf /tmp/prof_h; FSMG_XOV_HALF=3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o st -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_h -name "*.db" | head -1) 40 > $O/${TAG}_half3_timeline.txt 2>&1
tail -5 $O/${TAG}_pytest.log; cat $O/${TAG}_values.txt; head -12 $O/${TAG}_half3_timeline.txt
