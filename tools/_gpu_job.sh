# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): post-chain GEMMs side by side, A/B
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() { echo -n "$* : "; env "$@" python bench.py --config ${CFG:-cfg-B} --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d['spread']['ms_per_step_min'], d['spread']['ms_per_step_max'])"; }
for i in 1 2; do
run FSMG_POST_CONC=0
run FSMG_POST_CONC=1 FSMG_POST_SHARE=0
run FSMG_POST_CONC=1 FSMG_POST_SHARE=1
done
run FSMG_POST_CONC=0 FSMG_XCD_OVERLAP=0
run FSMG_POST_CONC=1 FSMG_POST_SHARE=0 FSMG_XCD_OVERLAP=0
run FSMG_POST_CONC=1 FSMG_POST_SHARE=1 FSMG_XCD_OVERLAP=0
for c in cfg-C cfg-D ref-default; do
CFG=$c run FSMG_POST_CONC=0
CFG=$c run FSMG_POST_CONC=1 FSMG_POST_SHARE=1
CFG=$c run FSMG_POST_CONC=1 FSMG_POST_SHARE=0
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x; FSMG_POST_CONC=1 FSMG_POST_SHARE=1 rocprofv3 --kernel-trace -d /tmp/prof_x -o st -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-breakdown > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_x -name "*.db" | head -1) 25 > $O/r04_post_conc_timeline.txt 2>&1
tail -30 $O/r04_post_conc_timeline.txt
