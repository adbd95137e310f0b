#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
HID=1024 BX3=1 VARIANT=1968 timeout 300 tools/xcd_chain_bench.bin 2>&1 | grep "vs CPU" | head -8
run() { local name=$1; local c=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$c $name', round(d['value'],1), round(d['ms_per_step'],4), 'fwd/bwd us', round(r.get('forward_us_per_time_step',0),2), round(r.get('backward_us_per_time_step',0),2))"
}
for rep in 1 2; do
run one_deep cfg-C FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_prev.so
run two_deep cfg-C FSMG_DUMMY=1
done
run one_deep cfg-C-T128 FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_prev.so
run two_deep cfg-C-T128 FSMG_DUMMY=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cfg-C or 1024 or pair16" 2>&1 | tail -3
