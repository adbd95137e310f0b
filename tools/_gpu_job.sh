cd $GRAFT_REPO_ROOT
for c in cfg-B cfg-C cfg-D cfg-E cfg-Bx8; do for m in 0 1; do FSMG_GEMM_H=$m timeout 600 python bench.py --config $c --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r03q_${c}_h$m.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/r03q_${c}_h$m.json').read().strip().splitlines()[-1])
ks=d.get('kernels') or {}
print('$c H=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], {k: round(v['ms_per_step'],4) for k,v in ks.items() if k.startswith('gemm')})
PY
done; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -2
