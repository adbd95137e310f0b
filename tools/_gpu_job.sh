cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_entry.py tests/test_gpu_parity.py -m gpu -x -q -k "maml_plugin_trains or library_owned" 2>&1 | tail -60 > gpurun_out/r03_pytest11.log; cat gpurun_out/r03_pytest11.log
