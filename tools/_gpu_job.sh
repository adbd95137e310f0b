#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_softmax" -p no:cacheprovider > $O/r05j_pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/r05j_pytest.log
cd /tmp
one() { l=$1; c=$2; shift; shift; env "$@" python $R/bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l $c', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d['final_loss'])"; }
for c in cfg-B cfg-D; do for r in 1 2; do one classic $c FSMG_FUSED_SOFTMAX=0; one fused $c FSMG_NOP=1; done; done
