#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_golden_g5.py tests/test_gpu_parity.py -q -m gpu -k "golden or frozen or full_size" -x 2>&1 | tail -25 > $O/r06_newtests.log
tail -8 $O/r06_newtests.log
timeout 900 python bench.py > $O/r06_bench_a.json 2> $O/r06_bench_a.err; tail -c 1500 $O/r06_bench_a.json; grep -i "fail\|error" $O/r06_bench_a.err | head
