# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): the round-end checks
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03s_pytest_final.log 2>&1; grep -n "passed\|failed\|Error\|assert" gpurun_out/r03s_pytest_final.log | head -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > gpurun_out/r03s_bench_final.json
python - <<PY
import json
d=json.load(open('gpurun_out/r03s_bench_final.json')); print('bench', round(d['value'],1), round(d['ms_per_step'],4), d['guard'], round(d['roofline']['frac'],4))
PY
