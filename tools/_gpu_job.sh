#!/bin/bash
# round-5 GPU job runner: one gpurun call = one box; everything writes under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05a}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/${TAG}_pytest.log
BASE=base:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0
timeout 200 python tools/ab_step.py $BASE all: inplace:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0 split:FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0 aside:FSMG_UPD_SPLIT=0,FSMG_INPLACE_DLOGITS=0 > $O/${TAG}_ab.log 2>&1
FSMG_CE_NT=0 timeout 200 python tools/ab_step.py $BASE all_nt0: inplace_nt0:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0 >> $O/${TAG}_ab.log 2>&1
for c in cfg-C cfg-E cfg-D ref-default; do echo "== $c" >> $O/${TAG}_ab.log; timeout 200 python tools/ab_step.py --config $c $BASE all: aside:FSMG_UPD_SPLIT=0,FSMG_INPLACE_DLOGITS=0 >> $O/${TAG}_ab.log 2>&1; done
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc $?"
tail -3 $O/${TAG}_pytest.log; cat $O/${TAG}_ab.log | grep -v "^\[" | tail -40
