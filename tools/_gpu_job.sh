# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in r03 new; do
  D=$R; [ $t = r03 ] && D=$R/_ab_r03
  (cd $D && python bench.py --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > $O/r04d_ab_${t}.json)
  rm -rf /tmp/prof_$t; (cd $D && rocprofv3 --kernel-trace --stats -d /tmp/prof_$t -o st -- python bench.py --no-cpu-baseline --no-breakdown > /dev/null 2>&1)
  python $R/tools/step_timeline.py $(find /tmp/prof_$t -name "*.db" | head -1) 200 > $O/r04d_step_timeline_$t.txt 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/prof_$t -name "*.db" | head -1) > $O/r04d_kernel_stats_$t.txt 2>&1
done
(cd $R && python bench.py --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > $O/r04d_ab_new2.json)
cd $R
python - <<PY
import json,glob
for n in sorted(glob.glob('gpurun_out/r04d_ab_*.json')):
    try:
        d=json.load(open(n)); print(n, round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], round(d['roofline']['frac'],4), round(d['roofline']['forward']['us_per_time_step'],3), round(d['roofline']['backward']['us_per_time_step'],3))
    except Exception as e:
        print(n, 'FAILED', e)
PY
cat $O/r04d_step_timeline_r03.txt; cat $O/r04d_step_timeline_new.txt
