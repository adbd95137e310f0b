#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05e}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "under_the_forward or maml_step_and_eval_match_oracle_at" -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
cd /tmp && export TMPDIR=/tmp
run() {  # name env...
  n=$1; shift
  for rep in 1 2; do env "$@" python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d['guard']['timeouts'])" >> $O/${TAG}_values.txt; done
}
run base FSMG_NOP=1
run ce_tail FSMG_CE_TAIL=1
run ce_tail_128 FSMG_CE_TAIL=1 FSMG_CE_TAIL_BLOCKS=128
run ce_tail_512 FSMG_CE_TAIL=1 FSMG_CE_TAIL_BLOCKS=512
run base FSMG_NOP=1
rm -rf /tmp/prof_ct; FSMG_CE_TAIL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ct -o st -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_ct -name "*.db" | head -1) 40 > $O/${TAG}_ce_tail_timeline.txt 2>&1
FSMG_CE_TAIL=1 python $R/bench.py --config cfg-D --steps 30 --warmup 8 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg-D ce_tail', round(d['value'],1), d['guard']['ok'])" >> $O/${TAG}_values.txt
python $R/bench.py --config cfg-D --steps 30 --warmup 8 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg-D base', round(d['value'],1), d['guard']['ok'])" >> $O/${TAG}_values.txt
tail -5 $O/${TAG}_pytest.log; cat $O/${TAG}_values.txt; cat $O/${TAG}_ce_tail_timeline.txt
