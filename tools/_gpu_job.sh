cd $GRAFT_REPO_ROOT
for c in cfg-C cfg-E cfg-B; do for m in 0 1; do FSMG_GEMM_WS=$m timeout 600 python bench.py --config $c --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r03e_${c}_ws$m.json 2> gpurun_out/r03e_${c}_ws$m.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03e_${c}_ws$m.json').read().strip().splitlines()[-1])
ks=d.get('kernels') or {}
print('$c WS=$m', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], {k: round(v['ms_per_step'],4) for k,v in ks.items() if k.startswith('gemm')})
PY
done; done
