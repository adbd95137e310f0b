cd $GRAFT_REPO_ROOT
( for d in 0 4 6; do for o in "logits  Hout" "dhout   dlog" "dW      (no"; do echo "== PROF BX3=2 DBG=$d $o"; DBG=$d PROF=1 BX3=2 ONLY="$o" timeout 120 tools/gemm_bench.bin 3 4 0 | grep -A3 "S 1 \|S 4 " | grep -v "CUs seen"; done; done ) > gpurun_out/r03_gemm_prof10.log 2>&1
cut -c1-250 gpurun_out/r03_gemm_prof10.log | grep "==\|PROF mult\|PROF load\| S "
