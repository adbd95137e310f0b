#!/bin/bash
# Counter passes of round 5: one counter set per rocprofv3 run, each under `timeout`, on the bare train loop
# (tools/pmc_workload.py) so that FETCH and WRITE describe the SAME kernel instantiations.  Run on the GPU box:
#   gpurun -- 'bash tools/pmc_passes.sh r05'        -> gpurun_out/r05_pmc*.{csv,txt,json}
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS=""
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  for fam in bx3 f32cell; do
    # bx3: the default handle (XCD-partitioned order: packed chains k_lstm_*_xcd16<4> until the serialised pair has timed out twice,
    # then k_lstm_*_xcd16<2>); f32cell: FSMG_XCD_OVERLAP=0 = the fp32 fused cell k_lstm_*_xcd<2> of the serial order
    rm -rf /tmp/pmc_${n}_$fam
    if [ $fam = f32cell ]; then export FSMG_XCD_OVERLAP=0; steps=3; else unset FSMG_XCD_OVERLAP; steps=6; fi
    export FSMG_AUX_TRIES=0      # no concurrency probe: under --pmc dispatches are serialised and every candidate stream would be rejected (-> serial order)
    timeout 240 rocprofv3 --pmc $set --kernel-trace -f csv -d /tmp/pmc_${n}_$fam -o p -- python $R/tools/pmc_workload.py $steps > $O/${TAG}_pmc_${n}_$fam.log 2>&1
    echo "pass $n $fam: exit $?" >> $O/${TAG}_pmc_passes.log
    f=$(find /tmp/pmc_${n}_$fam -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $O/${TAG}_pmc_${n}_$fam.csv; ARGS="$ARGS ${n}_$fam=$O/${TAG}_pmc_${n}_$fam.csv"; fi
  done
done
unset FSMG_XCD_OVERLAP FSMG_AUX_TRIES
python $R/tools/pmc_to_json.py $O/${TAG}_pmc.json $ARGS steps:bx3=6 steps:f32cell=3 > $O/${TAG}_pmc_summary.txt 2>&1
tail -5 $O/${TAG}_pmc_passes.log
