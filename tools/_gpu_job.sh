cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03j_pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r03j_pytest_gpu.log | tail -3
for c in cfg-C cfg-E cfg-D; do for m in 0 1; do FSMG_GEMM_BUF=$m timeout 600 python bench.py --config $c --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r03j_${c}_buf$m.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/r03j_${c}_buf$m.json').read().strip().splitlines()[-1])
print('$c BUF=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])
PY
done; done
