# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x -k "train_entry or maml or partitioned or forced_kernel or abi or unigram" > $O/r04q_pytest.log 2>&1; tail -15 $O/r04q_pytest.log
