#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/sanitize_run.sh
cp $O/san_ubsan.log $O/r06_sanitizer_ubsan_pytest.log; cp $O/san_asan_abi.log $O/r06_sanitizer_asan_abi_smoke.log
FSMG_XCD_BX3=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden_g5.py -q -m gpu 2>&1 | tail -8 > $O/r06_forced_xcd_bx3.log; tail -3 $O/r06_forced_xcd_bx3.log
FSMG_GEMM_H=2 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden_g5.py -q -m gpu 2>&1 | tail -8 > $O/r06_forced_gemm_h2.log; tail -3 $O/r06_forced_gemm_h2.log
