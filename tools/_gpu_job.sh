#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { local name=$1; local c=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c $name', round(d['value'],1), round(d['ms_per_step'],4), d['final_loss'])"
}
for rep in 1 2; do
for c in cfg-C cfg-E cfg-B; do
run prev $c FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_prev.so
run new $c FSMG_DUMMY=1
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gemm_variants.py -q -m gpu -x -k "bit or ident or full_size or trajectory" 2>&1 | tail -3
