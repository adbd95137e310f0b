#!/bin/bash
# round-5 GPU job runner: one gpurun call = one box; everything writes under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r05b}
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1
BASE=base:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0,FSMG_LAZY_CS=0
timeout 200 python tools/ab_step.py $BASE all: inplace:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0 split:FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0 aside:FSMG_UPD_SPLIT=0,FSMG_INPLACE_DLOGITS=0 base2:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0 > $O/${TAG}_ab.log 2>&1
FSMG_CE_NT=0 timeout 200 python tools/ab_step.py $BASE all_nt0: >> $O/${TAG}_ab.log 2>&1
for c in cfg-C cfg-E cfg-D ref-default; do echo "== $c" >> $O/${TAG}_ab.log; timeout 200 python tools/ab_step.py --config $c $BASE all: aside:FSMG_UPD_SPLIT=0,FSMG_INPLACE_DLOGITS=0,FSMG_LAZY_CS=0 split:FSMG_TAIL_ASIDE=0,FSMG_LAZY_CS=0 lazy:FSMG_TAIL_ASIDE=0,FSMG_UPD_SPLIT=0 >> $O/${TAG}_ab.log 2>&1; done
# cfg-E timeline (what is slow?)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_e; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o st -- python $R/bench.py --config cfg-E --steps 12 --warmup 4 --no-cpu-baseline --no-breakdown > $O/${TAG}_cfgE_bench.json 2> $O/${TAG}_cfgE_bench.err
python $R/tools/step_timeline.py $(find /tmp/prof_e -name "*.db" | head -1) 30 > $O/${TAG}_cfgE_step_timeline.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_e -name "*.db" | head -1) > $O/${TAG}_cfgE_kernel_stats.txt 2>&1
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lazy or split_update or inplace or selfcheck or maml" -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
grep -v "^\[\|Initializing\|amdgpu.ids" $O/${TAG}_ab.log | tail -60; tail -3 $O/${TAG}_pytest.log
