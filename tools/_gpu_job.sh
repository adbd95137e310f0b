cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -40 > gpurun_out/r03_pytest7.log; cat gpurun_out/r03_pytest7.log
timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r03_bench7.json 2> gpurun_out/r03_bench7.err; tail -3 gpurun_out/r03_bench7.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_bench7.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['guard']['ok'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('forward'), d.get('roofline',{}).get('backward'))
print({k:v['ms_per_step'] for k,v in d.get('kernels',{}).items()})
P
