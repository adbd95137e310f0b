#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cfg-C or cfg_C or 1024 or full_size or maml or every_tensor" 2>&1 | tail -8 > $O/p16_tests.log; tail -4 $O/p16_tests.log
for v in 0 1; do FSMG_XCD_BX3=$v timeout 300 python bench.py --config cfg-C --no-cpu-baseline --no-other-configs --no-extras --no-breakdown > $O/p16_bench_cfgC_bx3_$v.json 2>$O/p16_bench_cfgC_bx3_$v.err; python - <<PY
import json
d=json.loads(open('$O/p16_bench_cfgC_bx3_$v.json').read().strip().splitlines()[-1])
print('cfg-C FSMG_XCD_BX3=$v', d['value'], d['ms_per_step'], d.get('guard'))
PY
done
for v in 0 1; do FSMG_XCD_BX3=$v timeout 300 python bench.py --config cfg-C-T128 --no-cpu-baseline --no-other-configs --no-extras --no-breakdown > $O/p16_bench_cfgCT128_bx3_$v.json 2>$O/p16_bench_cfgCT128_bx3_$v.err; python - <<PY
import json
d=json.loads(open('$O/p16_bench_cfgCT128_bx3_$v.json').read().strip().splitlines()[-1])
print('cfg-C-T128 FSMG_XCD_BX3=$v', d['value'], d['ms_per_step'])
PY
done
