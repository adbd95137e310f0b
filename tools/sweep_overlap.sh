#!/bin/bash
# Sweep the two-stream schedule knobs on the cfg-B train step.
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.ljust(56), ': %.1f eps/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
run FSMG_AUX_BLOCKS=2 FSMG_NCHUNK=8
for cus in 28 24 20; do for ab in 2 3 4; do run FSMG_AUX_CUS=$cus FSMG_AUX_BLOCKS=$ab FSMG_NCHUNK=8; done; done
