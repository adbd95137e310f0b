cd $GRAFT_REPO_ROOT
( for bin in gemm_bench gemm_bench_pk gemm_bench gemm_bench_pk; do for o in "logits  Hout" "dhout   dlog" "dW      Hout" dKh edge; do for b in 1 2; do echo "== $bin BX3=$b $o"; BX3=$b ONLY="$o" timeout 120 tools/$bin.bin 10 4 1 | grep -v "verify.*ok" | grep "MISMATCH\|S 1 \|S 4 \|S 8 "; done; done; done ) > gpurun_out/r03_gemm_pk1.log 2>&1
grep -c MISMATCH gpurun_out/r03_gemm_pk1.log; grep -v edge gpurun_out/r03_gemm_pk1.log | grep "==\|S 1 .*logits\|S 4 \|S 8 " | cut -c1-190
