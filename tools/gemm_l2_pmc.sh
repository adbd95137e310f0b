#!/bin/bash
# L2 hit rate of one GEMM shape of tools/gemm_bench.bin under a tile order:  gpurun -- tools/gemm_l2_pmc.sh "logits  Hout" BX3 GROUP_M
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pl_$n; ONLY="$1" BX3=${2:-1} GROUP_M=${3:-0} rocprofv3 --pmc $set --kernel-trace -f csv -d /tmp/pl_$n -o p -- $R/tools/gemm_bench.bin 3 4 0 > /dev/null 2>&1
  f=$(find /tmp/pl_$n -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f 2>&1 | grep -v "k_ref\|reduce" | cut -c1-160
done
