cd $GRAFT_REPO_ROOT
( for o in "logits  Hout" "dhout   dlog" "dW      Hout" dKh zx edge; do for b in 2 3; do echo "== BX3=$b $o"; BX3=$b ONLY="$o" timeout 120 tools/gemm_bench.bin 10 4 1 | grep -v "verify.*ok"; done; done
for o in "logits  Hout" "dhout   dlog" "dW      (no"; do echo "== PROF BX3=3 $o"; PROF=1 BX3=3 ONLY="$o" timeout 120 tools/gemm_bench.bin 3 4 0 | grep -A3 "S 1 \|S 4 \|S 8 " | grep -v "CUs seen"; done ) > gpurun_out/r03_gemm_prof11.log 2>&1
grep -v edge gpurun_out/r03_gemm_prof11.log | grep "==\|S 1 \|S 3 \|S 4 \|S 5 \|S 8 \|PROF\|MISMATCH" | cut -c1-250
grep -c MISMATCH gpurun_out/r03_gemm_prof11.log
