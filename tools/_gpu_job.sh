#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_train_entry.py -q -m gpu -x -k "format_follows or maml or pair16 or cfg-C or 1024" 2>&1 | tail -8
for c in cfg-E cfg-C; do for rep in 1 2; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras --no-breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), round(d['ms_per_step'],4), d.get('guard',{}).get('ok'))"; done; done
FSMG_XCD_BX3=0 timeout 300 python bench.py --config cfg-E --no-cpu-baseline --no-other-configs --no-extras --no-breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg-E forced fp32 chains', round(d['value'],1), round(d['ms_per_step'],4))"
