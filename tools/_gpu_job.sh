#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { local name=$1; local c=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$c $name', round(d['value'],1), round(d['ms_per_step'],4), 'fwd/bwd us', round(r.get('forward_us_per_time_step',0),2), round(r.get('backward_us_per_time_step',0),2))"
}
for rep in 1 2; do
run q0 ref-default FSMG_XCD_VARIANT=48 FSMG_XCD_VARIANT_BWD=32
run q2 ref-default FSMG_XCD_VARIANT=16432 FSMG_XCD_VARIANT_BWD=32
run q4 ref-default FSMG_XCD_VARIANT=32816 FSMG_XCD_VARIANT_BWD=32
run q6 ref-default FSMG_XCD_VARIANT=49200 FSMG_XCD_VARIANT_BWD=32
done
