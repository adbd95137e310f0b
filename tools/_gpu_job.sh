cd $GRAFT_REPO_ROOT
FSMG_GEMM_H=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gemm_variants.py -m gpu -x -q > gpurun_out/r03s_pytest_h2.log 2>&1
tail -3 gpurun_out/r03s_pytest_h2.log
