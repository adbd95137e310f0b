# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): the fused cell alone and in the serial-order step, one box
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do PIPE=4 HID=512 timeout 300 tools/xcd_chain_bench.bin > $O/r04_cell_standalone_$i.log 2>&1; done
grep -n "\[4\]" $O/r04_cell_standalone_1.log | head -40
for i in 1 2; do FSMG_XCD_OVERLAP=0 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_cell_in_serial_step_$i.json; python -c "
import json; d=json.load(open('$O/r04_cell_in_serial_step_$i.json')); r=d['roofline']; print('serial order: fwd %.3f bwd %.3f us/step, clock %.3f GHz, frac %.4f, value %.1f' % (r['forward']['us_per_time_step'], r['backward']['us_per_time_step'], r['clock_ghz'], r['frac'], d['value']))"; done
