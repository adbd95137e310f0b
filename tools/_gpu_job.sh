cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "1 1" "1 0" "0 0"; do
  set -- $c
  rm -rf /tmp/p_$1$2; FSMG_LOGITS_NT=$1 FSMG_DLOGITS_NT=$2 rocprofv3 --kernel-trace --stats -d /tmp/p_$1$2 -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r03t_ntp_$1$2.json
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p_$1$2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03t_ntp_$1$2_stats.txt 2>&1
  echo "== logits_nt=$1 dlogits_nt=$2"; grep "k_ce_rows_reg\|k_gemm_bx3h" $GRAFT_REPO_ROOT/gpurun_out/r03t_ntp_$1$2_stats.txt | cut -c1-60,100-160
  python - <<PY
import json
d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/r03t_ntp_$1$2.json')); print(round(d['value'],1))
PY
done
