#!/bin/bash
# PMC pass over one GEMM shape of tools/gemm_bench.bin:  gpurun -- tools/gemm_pmc.sh logits [BX3]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pg_$n; ONLY="$1" BX3=${2:-0} rocprofv3 --pmc $set --kernel-trace -f csv -d /tmp/pg_$n -o p -- $R/tools/gemm_bench.bin 3 4 0 > /dev/null 2>&1
  f=$(find /tmp/pg_$n -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f 2>&1 | grep -v "k_ref\|reduce" | cut -c1-200
done
