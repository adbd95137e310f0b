cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r03g_pytest_gpu.log | tail -3
for c in cfg-B cfg-C cfg-E; do timeout 600 python bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r03g_$c.json 2> gpurun_out/r03g_$c.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03g_$c.json').read().strip().splitlines()[-1])
print('$c', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])
PY
done
