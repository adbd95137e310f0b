run() { echo -n "$* : "; env "$@" FSMG_XCD_OVERLAP=1 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])"; }
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=15
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=20
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=3 FSMG_XOV_DW_SHARE=20
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=4 FSMG_XOV_DW_SHARE=20
run FSMG_XOV_DW_SPLIT=6 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=15
run FSMG_XOV_DW_SPLIT=12 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=15
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=10
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=0
run FSMG_XOV_DW_SPLIT=8 FSMG_XOV_BLOCKS=2 FSMG_XOV_DW_SHARE=0 FSMG_XOV_HEAD=1
echo -n "baseline: "; python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"
