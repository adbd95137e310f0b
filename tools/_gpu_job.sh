cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest_gpu.log 2>&1
tail -4 gpurun_out/r03f_pytest_gpu.log
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
print('cfg-B', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])
PY
FSMG_CHAIN_SPIN_LIMIT=0 timeout 300 python - <<PY
print('skip')
PY
