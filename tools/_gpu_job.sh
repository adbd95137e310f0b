#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05c}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
run() {  # name env...
  n=$1; shift
  rm -rf /tmp/prof_$n
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o st -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs > $O/${TAG}_${n}_bench.json 2> $O/${TAG}_${n}_bench.err
  python $R/tools/step_timeline.py $(find /tmp/prof_$n -name "*.db" | head -1) 40 > $O/${TAG}_${n}_timeline.txt 2>&1
  env "$@" python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], d['guard']['ok'])" >> $O/${TAG}_values.txt
}
run all FSMG_NOP=1
run base FSMG_UPD_SPLIT=0 FSMG_TAIL_ASIDE=0 FSMG_INPLACE_DLOGITS=0
run aside FSMG_UPD_SPLIT=0 FSMG_INPLACE_DLOGITS=0
run aside_inplace FSMG_UPD_SPLIT=0
run aside_split FSMG_INPLACE_DLOGITS=0
cat $O/${TAG}_values.txt
