cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_pytest12.log; cat gpurun_out/r03_pytest12.log
for c in cfg-C cfg-E cfg-D cfg-Bx8; do timeout 200 python bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline 2> gpurun_out/r03_bench12_$c.err | tail -1 > gpurun_out/r03_bench12_$c.json; python - $c <<'P'
import json,sys
d=json.loads(open('gpurun_out/r03_bench12_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d.get('roofline',{}).get('frac'), {k:round(v['ms_per_step'],4) for k,v in d.get('kernels',{}).items() if 'lstm' in k})
P
done
