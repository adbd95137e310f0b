# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): round-end checks, then the artefacts
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/r04x_pytest.log 2>&1; grep -n "passed\|failed" $O/r04x_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/refresh_profiles.sh r04 > $O/r04_refresh.log 2>&1
python - <<PY
import json,glob
for n in sorted(glob.glob('gpurun_out/r04_bench*.json')):
    try:
        d=json.loads(open(n).read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('%-50s' % n[11:], round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], 'frac', round(r.get('frac',0),4), r.get('schedule'), r.get('clock_ghz'), round((d.get('roofline_step') or {}).get('frac',0),3))
    except Exception as e:
        print(n, 'FAILED', e)
PY
