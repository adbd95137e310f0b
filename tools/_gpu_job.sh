cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r03 > gpurun_out/r03_refresh.log 2>&1
python - <<PY
import json
for f in ['r03_bench.json','r03_bench_under_rocprofv3.json','r03_bench_cfg-C.json','r03_bench_cfg-E.json','r03_bench_cfg-D.json','r03_bench_cfg-Bx8.json']:
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1])
    print(f, round(d['value'],1), round(d['ms_per_step'],4), d.get('roofline',{}).get('frac'), d['guard']['ok'], (d.get('roofline_gemm') or {}).get('frac'))
PY
