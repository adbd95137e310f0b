# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): the round-end checks
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/r04s_pytest.log 2>&1; tail -12 $O/r04s_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > $O/r04s_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04s_bench.json')); print('bench', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], round(d['roofline']['frac'],4), d['roofline'].get('schedule'), round(d['roofline_step']['frac'],3), d['cpu_baseline']['value'], d['extras_failed'])
PY
