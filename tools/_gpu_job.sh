cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest_gpu.log 2>&1
tail -15 gpurun_out/r03b_pytest_gpu.log
for m in 1 0; do FSMG_GEMM_WS=$m timeout 600 python bench.py --steps 40 --warmup 8 > gpurun_out/r03c_bench_ws$m.json 2> gpurun_out/r03c_bench_ws$m.err; tail -c 400 gpurun_out/r03c_bench_ws$m.json | head -c 10; python - <<PY
import json
d=json.loads(open('gpurun_out/r03c_bench_ws$m.json').read().strip().splitlines()[-1])
print('WS=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])
PY
done
