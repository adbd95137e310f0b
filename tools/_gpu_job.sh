#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05d}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc $?"
bash tools/refresh_profiles.sh $TAG
tail -3 $O/${TAG}_pytest.log; tail -2 $O/${TAG}_smoke.log; cat $O/${TAG}_pmc_passes.log; head -30 $O/${TAG}_pmc_summary.txt
python - <<PY
import json
for n in ('bench', 'bench_serial_order', 'bench_cfg-C', 'bench_cfg-D', 'bench_cfg-E', 'bench_ref-default', 'bench_cfg-Bx8', 'bench_same_gpu_4ranks'):
    try:
        d = json.load(open('$O/${TAG}_%s.json' % n)); print(n, round(d['value'], 1), round(d['ms_per_step'], 4), d['guard']['ok'], d.get('extras_failed'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
