# scratch job script: one test
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "change_from_call_to_call" 2>&1 | tail -30
