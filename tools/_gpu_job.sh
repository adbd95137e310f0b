#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
for rep in 1 2; do timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > $O/${TAG}_pytest_$rep.log 2>&1; echo "pytest $rep rc $?"; tail -1 $O/${TAG}_pytest_$rep.log; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc $?"
bash tools/refresh_profiles.sh $TAG
cat $O/${TAG}_pmc_passes.log
python - <<PY
import json
for n in ('bench', 'bench_serial_order', 'bench_cfg-C', 'bench_cfg-D', 'bench_cfg-E', 'bench_ref-default', 'bench_cfg-Bx8', 'bench_same_gpu_4ranks'):
    try:
        d = json.load(open('$O/${TAG}_%s.json' % n)); print(n, round(d['value'], 1), round(d['ms_per_step'], 4), d['guard']['ok'], d.get('extras_failed'))
    except Exception as e:
        print(n, 'FAILED', e)
d = json.load(open('$O/${TAG}_bench.json'))
print({k: (round(v['value'], 1), v['guard']['ok']) for k, v in d['other_configs'].items()}, d['other_configs']['cfg-B-serial-order']['fused_cell']['frac'])
print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel_variant'], d['roofline_step']['frac'], d['roofline_gemm']['frac'])
PY
