# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python bench.py --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > $O/r04w_bench.json
python bench.py --config ref-default --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > $O/r04w_refdefault.json
python - <<PY
import json
for n in ('r04w_bench','r04w_refdefault'):
    d=json.load(open('gpurun_out/%s.json'%n)); r=d['roofline']; print(n, round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], r.get('clock_ghz'))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split or every_tensor or cut_points or graph_replay" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rd; rocprofv3 --kernel-trace --stats -d /tmp/prof_rd -o st -- python $R/bench.py --config ref-default --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > /dev/null
python $R/tools/step_timeline.py $(find /tmp/prof_rd -name "*.db" | head -1) 150 2>&1 | grep "multi_op\|span"
