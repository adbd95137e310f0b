#!/bin/bash
# scratch runner for one gpurun call (edit, then: gpurun -- 'bash tools/_gpu_job.sh'); what it leaves under gpurun_out/ comes back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused_softmax" 2>&1 | tail -5
for c in ref-default cfg-C cfg-E; do for v in 1 0 1 0; do echo -n "$c fused=$v: "; FSMG_BENCH_REPEATS=3 FSMG_FUSED_SOFTMAX=$v timeout 300 python bench.py --config $c --steps 30 --warmup 8 --no-cpu-baseline --no-breakdown --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['guard'].get('fused_softmax_taken'), d['guard']['ok'])"; done; done
