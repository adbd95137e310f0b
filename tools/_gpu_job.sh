#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r06_gemm_planes1.log; : > $L
for pl in 0 2 3 1; do
  for sh in "logits  Hout" "dhout   dlog"; do
    echo "=== PL=$pl $sh" >> $L
    BX3=3 PL=$pl ONLY="$sh" timeout 300 tools/gemm_bench.bin 20 4 0 >> $L 2>&1
  done
done
echo "=== edges + verify" >> $L
for pl in 2 3 1; do
  BX3=3 PL=$pl ONLY="edge" timeout 300 tools/gemm_bench.bin 2 4 1 >> $L 2>&1
  BX3=3 PL=$pl ONLY="dW      (no" timeout 300 tools/gemm_bench.bin 5 4 1 >> $L 2>&1
done
BX3=3 PL=3 ONLY="logits  Hout" timeout 300 tools/gemm_bench.bin 3 4 1 >> $L 2>&1
BX3=3 PL=2 DIST=1 ONLY="dhout   dlog" timeout 300 tools/gemm_bench.bin 3 4 1 >> $L 2>&1
tail -5 $L
