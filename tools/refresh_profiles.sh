#!/bin/bash
# Regenerates the round artefacts on the GPU box into gpurun_out/ (copy the ones to keep into profiles/).
#   gpurun -- 'bash tools/refresh_profiles.sh r06'
# Every profiler invocation runs under `timeout` (round 4, call 47: a --pmc pass never returned); the counter passes live in
# tools/pmc_passes.sh (one counter set per run, on the bare train loop).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json
rm -rf /tmp/prof_st; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_st -o st -- python $R/bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprofv3.json
python $R/tools/rocpd_stats.py $(find /tmp/prof_st -name "*.db" | head -1) > $O/${TAG}_bench_rocprofv3_kernel_stats.txt 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_st -name "*.db" | head -1) 40 > $O/${TAG}_step_timeline.txt 2>&1
# the other BASELINE.json configurations: bench line (un-profiled), per-kernel statistics and device timeline (profiled)
for c in cfg-C cfg-C-T128 cfg-E cfg-D ref-default; do
  t=$(echo $c | tr -d '-')
  python $R/bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_$c.json
  rm -rf /tmp/prof_$t; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$t -o st -- python $R/bench.py --config $c --steps 20 --warmup 6 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 > /dev/null
  python $R/tools/rocpd_stats.py $(find /tmp/prof_$t -name "*.db" | head -1) > $O/${TAG}_${t}_rocprofv3_kernel_stats.txt 2>&1
  python $R/tools/step_timeline.py $(find /tmp/prof_$t -name "*.db" | head -1) 40 > $O/${TAG}_${t}_step_timeline.txt 2>&1
done
python $R/bench.py --config cfg-Bx8 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg-Bx8.json
# the serial order beside the partitioned one (same box)
FSMG_XCD_OVERLAP=0 python $R/bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/${TAG}_bench_serial_order.json
# N > 1 dry run on ONE GPU over gloo (launcher, exchange schedules, guards; the numbers mean nothing: four processes share the chip)
FSMG_BENCH_SAME_GPU=1 FSMG_BENCH_REPEATS=2 timeout 600 python $R/bench.py --gpus 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_same_gpu_4ranks.json 2> $O/${TAG}_bench_same_gpu_4ranks.err
bash $R/tools/pmc_passes.sh $TAG
