#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like_bench.json 2> $O/driver_like_bench.err ) 2>&1 | tail -3
tail -1 $O/driver_like_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')}); print(d['config']['workload'] if 'workload' in d['config'] else d['config']); print({k:d['roofline'][k] for k in ('bound','achieved','peak','unit','frac','traffic')}); print({k:d['cpu_baseline'][k] for k in ('value','unit','cores','kind')}); print(d.get('extras_failed'))"
wc -l $O/driver_like_bench.json
