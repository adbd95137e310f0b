cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_gemm_hq1.log; : > $O
for x in 3 -1 0; do
echo "=== BX3=3 XCD_FIRST=$x" >> $O
BX3=3 XCD_FIRST=$x ONLY="d" timeout 300 tools/gemm_bench.bin 10 4 1 >> $O 2>&1; echo "rc $?" >> $O
done
grep -c " ok" $O; grep -c MISMATCH $O
grep "===\|^dW \|^dhout\|^logits\|rc " $O | grep "===\|S 3 \|S 5 \|S 6 \|S 8 \|rc " | cut -c1-200
