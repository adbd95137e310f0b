cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 4 > gpurun_out/r03m_bench.json 2> gpurun_out/r03m_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03m_bench.json').read().strip().splitlines()[-1])
print('cfg-B', round(d['value'],1), d['guard']['ok'], d['extras_failed'], sorted(d.keys()))
PY
timeout 900 python -m pytest tests/test_bench_launch.py -m gpu -x -q 2>&1 | grep "passed\|failed"
