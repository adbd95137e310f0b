#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cfg-C or cfg_C or full_size or 1024" 2>&1 | tail -15 > $O/p16_tests_xov.log; tail -6 $O/p16_tests_xov.log
run() { # name, env...
  local name=$1; shift
  for c in cfg-C cfg-C-T128; do env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras --no-breakdown > $O/p16_bench_${c}_$name.json 2>$O/p16_bench_${c}_$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/p16_bench_${c}_$name.json').read().strip().splitlines()[-1])
    print('$c $name', round(d['value'],1), round(d['ms_per_step'],4), d.get('guard',{}).get('ok'), d.get('guard',{}).get('timeouts'))
except Exception as e: print('$c $name failed', e); print(open('$O/p16_bench_${c}_$name.err').read()[-1500:])
PY
  done
}
run serial FSMG_XCD_OVERLAP=0

run parts6 FSMG_XOV_PARTS=6


cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o st -- python $R/bench.py --config cfg-C --steps 20 --warmup 6 --no-cpu-baseline --no-breakdown --no-other-configs --no-extras 2>/dev/null | tail -1 > /dev/null
python $R/tools/step_timeline.py $(find /tmp/prof_c -name "*.db" | head -1) 40 > $O/p16_cfgC_xov_step_timeline.txt 2>&1
head -45 $O/p16_cfgC_xov_step_timeline.txt
