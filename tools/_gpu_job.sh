#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
cd /tmp; s=$(date +%s); python $R/bench.py > $O/r05h_bench.json 2> $O/r05h_bench.err; e=$(date +%s); echo "bench wall $((e-s)) s rc $?"
grep other_configs $O/r05h_bench.err
python - <<PY
import json
d = json.load(open('$O/r05h_bench.json'))
print(round(d['value'],1), d['guard']['ok'], d['extras_failed'])
print({k: (round(v.get('value', 0), 1), v.get('guard', {}).get('ok'), round(v.get('leg_s', 0), 1)) for k, v in d['other_configs'].items()}, d['other_configs']['cfg-B-serial-order'].get('fused_cell', {}).get('frac'))
PY
