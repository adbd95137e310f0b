#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cfg-C or 1024 or pair16 or partitioned or forced or format" 2>&1 | tail -3
for rep in 1 2; do for c in cfg-C cfg-C-T128; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$c', round(d['value'],1), round(d['ms_per_step'],4), 'fwd/bwd us', round(r.get('forward_us_per_time_step',0),2), round(r.get('backward_us_per_time_step',0),2), 'cell', round(r.get('frac',0),3))"; done; done
