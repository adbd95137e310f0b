#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/probe_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "aside or fused_softmax or split_update" 2>&1 | tail -3
for c in ref-default cfg-C cfg-E cfg-D cfg-B; do for v in 1 0 1 0; do echo -n "$c tail_aside=$v: "; FSMG_BENCH_REPEATS=3 FSMG_TAIL_ASIDE=$v timeout 300 python bench.py --config $c --steps 30 --warmup 8 --no-cpu-baseline --no-breakdown --no-other-configs --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; done; done
