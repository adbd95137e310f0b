# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): the round-end checks
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/r04x_pytest.log 2>&1; grep -n "passed\|failed" $O/r04x_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r04x_bench.json 2> $O/r04x_bench.err; wc -l < $O/r04x_bench.json; python -c "
import json; d=json.loads(open('$O/r04x_bench.json').read()); print(d['value'], d['ms_per_step'], d['guard']['ok'], d.get('extras_failed'))"
