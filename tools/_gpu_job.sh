cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_tensor or full_size" > gpurun_out/r03s_pytest_auto.log 2>&1; grep -n "passed\|failed\|Error\|assert" gpurun_out/r03s_pytest_auto.log | head -8
python bench.py --config cfg-D --steps 40 --warmup 8 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > gpurun_out/r03s_cfgD_auto.json
python - <<PY
import json
d=json.load(open('gpurun_out/r03s_cfgD_auto.json')); print('cfgD auto', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['roofline']['kernel'][:80])
PY
