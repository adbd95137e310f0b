cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_backward or library_owned" > gpurun_out/t1.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/t1.log | tail -5
timeout 900 python -m pytest tests/test_bench_launch.py -m gpu -x -q > gpurun_out/t2.log 2>&1; tail -40 gpurun_out/t2.log
