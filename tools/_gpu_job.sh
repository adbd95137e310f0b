cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03n_pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r03n_pytest_gpu.log | tail -3; grep -n "Error\|assert " gpurun_out/r03n_pytest_gpu.log | head -8
for m in 0 1 0 1; do FSMG_DW_TRANSPOSE=$m timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r03n_bench_tr$m.json 2> gpurun_out/r03n_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03n_bench_tr$m.json').read().strip().splitlines()[-1])
ks=d.get('kernels') or {}
print('cfg-B TR=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], {k: round(v['ms_per_step'],4) for k,v in ks.items() if k.startswith('gemm')})
PY
done
for c in cfg-C cfg-D; do for m in 0 1; do FSMG_DW_TRANSPOSE=$m timeout 600 python bench.py --config $c --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r03n_${c}_tr$m.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/r03n_${c}_tr$m.json').read().strip().splitlines()[-1])
ks=d.get('kernels') or {}
print('$c TR=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], {k: round(v['ms_per_step'],4) for k,v in ks.items() if k.startswith('gemm_dw')})
PY
done; done
