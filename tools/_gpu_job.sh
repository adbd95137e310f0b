#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden_g5.py -q -m gpu -k "full_size_gradients or selfcheck or fused_softmax_falls or timeout or frozen" 2>&1 | tail -15 > $O/r06_gputests_c.log; tail -4 $O/r06_gputests_c.log
python tools/ab_step.py --steps 40 --rounds 5 base: nt:FSMG_BWD_NT=1 rev:FSMG_DH_REV=1 both:FSMG_BWD_NT=1,FSMG_DH_REV=1 > $O/r06_ab_nt_rev.txt 2>&1; tail -9 $O/r06_ab_nt_rev.txt
cd /tmp && export TMPDIR=/tmp
export FSMG_AUX_TRIES=0
for nt in 0 1; do
  for set in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pmc_$set_$nt
    FSMG_BWD_NT=$nt FSMG_DH_REV=$nt timeout 240 rocprofv3 --pmc $set --kernel-trace -f csv -d /tmp/pmc_${set}_$nt -o p -- python $R/tools/pmc_workload.py 6 > $O/r06_pmc_${set}_nt$nt.log 2>&1
    f=$(find /tmp/pmc_${set}_$nt -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $O/r06_pmc_${set}_nt$nt.csv
  done
  python $R/tools/pmc_to_json.py $O/r06_pmc_nt$nt.json WRITE_SIZE_bx3=$O/r06_pmc_WRITE_SIZE_nt$nt.csv FETCH_SIZE_bx3=$O/r06_pmc_FETCH_SIZE_nt$nt.csv steps:bx3=6 > $O/r06_pmc_nt${nt}_summary.txt 2>&1
done
grep -h "k_lstm_bwd_xcd16\|k_gemm_bx3h<0, 0" $O/r06_pmc_nt0_summary.txt $O/r06_pmc_nt1_summary.txt
