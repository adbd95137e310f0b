# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): the round-end checks + ref-default artefacts
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/r04u_pytest.log 2>&1; tail -6 $O/r04u_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --config ref-default --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_ref-default.json
FSMG_XCD=0 python $R/bench.py --config ref-default --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_ref-default_column_split.json
FSMG_HP_ALIGN=16 python $R/bench.py --config ref-default --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_ref-default_hp208.json
rm -rf /tmp/prof_rd; rocprofv3 --kernel-trace --stats -d /tmp/prof_rd -o st -- python $R/bench.py --config ref-default --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > /dev/null
python $R/tools/step_timeline.py $(find /tmp/prof_rd -name "*.db" | head -1) 150 > $O/r04_refdefault_step_timeline.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_rd -name "*.db" | head -1) > $O/r04_refdefault_rocprofv3_kernel_stats.txt 2>&1
cat $O/r04_refdefault_step_timeline.txt
cd $R
python - <<PY
import json,glob
for n in sorted(glob.glob('gpurun_out/r04_bench_ref-default*.json')):
    d=json.load(open(n)); print('%-50s' % n[11:], round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], 'fwd', round(d['roofline']['forward']['us_per_time_step'],3), 'bwd', round(d['roofline']['backward']['us_per_time_step'],3), 'frac', round(d['roofline']['frac'],4), round(d['roofline_step']['frac'],3))
PY
