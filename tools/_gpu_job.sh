# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment)
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1
tail -5 gpurun_out/r04_refresh.log
python - <<PY
import json,glob
for n in sorted(glob.glob('gpurun_out/r04_bench*.json')):
    try:
        d=json.load(open(n)); print('%-50s' % n[11:], round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], 'frac', round(d.get('roofline',{}).get('frac',0),4), d.get('roofline',{}).get('schedule'), round((d.get('roofline_step') or {}).get('frac',0),3))
    except Exception as e:
        print(n, 'FAILED', e)
PY
