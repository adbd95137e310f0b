# scratch job script for `gpurun -- 'bash tools/_gpu_job.sh'` (overwritten per experiment): merged dK GEMM
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gemm_variants.py -m gpu -q -x > $O/r04x_pytest_a.log 2>&1; tail -3 $O/r04x_pytest_a.log | head -2
run() { echo -n "${CFG:-cfg-B} $* : "; env "$@" python bench.py --config ${CFG:-cfg-B} --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], round(d['final_loss'],6))"; }
for i in 1 2; do run FSMG_MERGE_DK=0; run FSMG_MERGE_DK=1; done
for c in cfg-C cfg-D cfg-E ref-default; do CFG=$c run FSMG_MERGE_DK=0; CFG=$c run FSMG_MERGE_DK=1; done
timeout 2500 python -m pytest tests -m gpu -q -x > $O/r04x_pytest.log 2>&1; tail -4 $O/r04x_pytest.log | head -2
