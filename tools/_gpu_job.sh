#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > $O/r05i_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/r05i_pytest.log
python $R/tools/probe_handle_ordinal.py own 2>/dev/null | grep handle
V="h1: h2: h3: h4: h5: h6: h7:"; timeout 300 python tools/ab_step.py --rounds 3 --steps 20 $V 2>/dev/null | grep "episodes/s" | cut -c1-90
cd /tmp; python $R/bench.py 2>/dev/null | tail -1 > $O/r05i_bench.json; python -c "
import json; d = json.load(open('$O/r05i_bench.json')); print(round(d['value'],1), d['guard']['ok'], {k: round(v['value'],1) for k, v in d['other_configs'].items()})"
