#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
P=$R/few-shot-music-generation_amd/lib/libfsmg_prev.so
for v in prev new0 prev new0 prev new0 new24; do
  echo -n "$v: "
  if [ $v = prev ]; then export FSMG_LIB=$P; unset FSMG_ZX_HEAD; elif [ $v = new0 ]; then unset FSMG_LIB; export FSMG_ZX_HEAD=0; else unset FSMG_LIB; export FSMG_ZX_HEAD=24; fi
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'])"; done
