# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/r04o_pytest.log 2>&1; tail -25 $O/r04o_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
