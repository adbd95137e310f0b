#!/bin/bash
# Regenerates the round artefacts on the GPU box into gpurun_out/ (copy the ones to keep into profiles/).
#   gpurun -- tools/refresh_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r_bench.json 2> $O/r_bench.err
FSMG_OVERLAP=0 python $R/bench.py --no-cpu-baseline > $O/r_bench_single.json 2>/dev/null
rm -rf /tmp/prof_st; rocprofv3 --kernel-trace --stats -d /tmp/prof_st -o st -- python $R/bench.py --no-cpu-baseline > $O/r_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/prof_st -name "*.db" | head -1) > $O/r_kernel_stats.txt 2>&1
python $R/tools/step_timeline.py $(find /tmp/prof_st -name "*.db" | head -1) 30 > $O/r_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c; rocprofv3 --pmc $c --kernel-trace -f csv -d /tmp/prof_$c -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown > /dev/null 2>&1
  python - $(find /tmp/prof_$c -name "*counter_collection.csv" | head -1) $c >> $O/r_pmc_traffic.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_gemm' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[2]:
        agg[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'])].append(float(r['Counter_Value']))
for (k, g), v in sorted(agg.items()):
    print('%s %-40s grid %-9s launches %3d mean %.3f' % (sys.argv[2], k, g, len(v), sum(v) / len(v)))
PY
done
