cd $GRAFT_REPO_ROOT
timeout 300 tools/xcd_chain_bench.bin > gpurun_out/r03_xcd_probe7.log 2>&1; grep -i "us per\|per step\|max err\|FAIL\|mismatch" gpurun_out/r03_xcd_probe7.log | head -30
HID=1024 timeout 300 tools/xcd_chain_bench.bin > gpurun_out/r03_pair_probe7.log 2>&1; grep -i "us per\|per step\|max err\|FAIL\|mismatch" gpurun_out/r03_pair_probe7.log | head -20
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03l_pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r03l_pytest_gpu.log | tail -3; grep -n "Error\|assert" gpurun_out/r03l_pytest_gpu.log | head -10
for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03l_bench.json').read().strip().splitlines()[-1])
print('cfg-B', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], d['roofline']['frac'], d['roofline']['forward']['us_per_time_step'], d['roofline']['backward']['us_per_time_step'])
PY
done
