#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { # name, config, env...
  local name=$1; local c=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras --no-breakdown > $O/ab_${c}_$name.json 2>$O/ab_${c}_$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/ab_${c}_$name.json').read().strip().splitlines()[-1])
    print('$c $name', round(d['value'],1), round(d['ms_per_step'],4), d.get('guard',{}).get('ok'), d.get('guard',{}).get('timeouts'))
except Exception as e: print('$c $name failed', e); print(open('$O/ab_${c}_$name.err').read()[-1500:])
PY
}
for rep in 1 2; do
run old$rep cfg-B FSMG_XCD_VARIANT=32
run new$rep cfg-B FSMG_XCD_VARIANT=2080
run strm$rep cfg-B FSMG_XCD_VARIANT=2768
done
run old cfg-D FSMG_XCD_VARIANT=32
run new cfg-D FSMG_XCD_VARIANT=2080
run old2 cfg-D FSMG_XCD_VARIANT=32

timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "forced" 2>&1 | tail -5
