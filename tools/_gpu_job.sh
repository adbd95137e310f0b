cd $GRAFT_REPO_ROOT
( for o in "logits  Hout" "dhout   dlog" "dW      Hout" dKh zx edge "logits/8"; do for b in 1 2; do echo "== BX3=$b $o"; BX3=$b ONLY="$o" timeout 120 tools/gemm_bench.bin 10 4 1 | grep -v "verify.*ok" | grep "MISMATCH\|S 1 \|S 4 \|S 8 "; done; done ) > gpurun_out/r03_gemm_buf4.log 2>&1
grep -c MISMATCH gpurun_out/r03_gemm_buf4.log; grep -v edge gpurun_out/r03_gemm_buf4.log | grep "==\|S 1 .*logits\|S 1 .*zx\|S 4 \|S 8 " | cut -c1-200 | tail -40
for m in 1 1; do timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r03k_bench.json 2> gpurun_out/r03k_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03k_bench.json').read().strip().splitlines()[-1])
ks=d.get('kernels') or {}
print('cfg-B', round(d['value'],1), round(d['ms_per_step'],4), d['guard']['ok'], {k: round(v['ms_per_step'],4) for k,v in ks.items() if k.startswith('gemm')})
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -2
