#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (one per counter set, tools/pmc_passes.sh) -> one JSON: per kernel INSTANTIATION the mean counter values,
launches and mean duration, HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KB as rocprofv3 reports them, no factor: DESIGN.md 9.5)
and the matrix-pipe utilisation SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs).

  python tools/pmc_to_json.py out.json label=csv [label=csv ...]
"""
import collections
import csv
import json
import re
import sys

CLOCK_HZ = 2.4e9          # spec clock; under --pmc the chip idles between serialised dispatches, so the sustained clock is not lower
SIMDS = 256 * 4
out_path, pairs = sys.argv[1], [a.split('=', 1) for a in sys.argv[2:] if not a.startswith('steps:')]
step_counts = {a.split(':', 1)[1].split('=')[0]: int(a.split('=')[1]) for a in sys.argv[2:] if a.startswith('steps:')}      # steps:<family>=n
totals = collections.defaultdict(float)
kern = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for label, path in pairs:
    seen = set()
    for r in csv.DictReader(open(path)):
        name = re.sub(r'fsmg::\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name).replace('void ', '').strip()
        kern[name][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE') and (name.startswith('k_') or name.startswith('k_', name.find(' ') + 1)):
            totals[label] += 1024.0 * float(r['Counter_Value'])
        key = (label, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            dur[name].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
res = {}
for name, cs in kern.items():
    if not (name.startswith('k_') or 'k_gemm' in name or 'k_lstm' in name or 'k_ce' in name):
        continue
    e = {'launches': max(len(v) for v in cs.values()), 'mean_duration_us': sum(dur[name]) / max(len(dur[name]), 1) / 1e3}
    for c, v in cs.items():
        e[c + ('_KB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else '')] = sum(v) / len(v)
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        e['hbm_bytes_per_launch'] = 1024.0 * (e['FETCH_SIZE_KB'] + e['WRITE_SIZE_KB'])
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and e['mean_duration_us'] > 0:
        e['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['mean_duration_us'] * 1e-6 * CLOCK_HZ * SIMDS)
    res[name] = e
doc = {'source': 'rocprofv3 --pmc <one counter set per pass> --kernel-trace -- python tools/pmc_workload.py (tools/pmc_passes.sh); means per launch, per kernel instantiation',
       'workload': 'cfg-B train steps (tools/pmc_workload.py)',
       'units': {'FETCH_SIZE_KB / WRITE_SIZE_KB': 'KB as rocprofv3 reports them, raw (no x2: DESIGN.md 9.5 calibrates that factor for these access patterns)',
                 'mfma_busy_frac': 'SQ_VALU_MFMA_BUSY_CYCLES / (mean duration x 2.4 GHz x 1024 SIMDs); durations are those of the SERIALISED dispatches of the counter pass'},
       'passes': {label: path.split('/')[-1] for label, path in pairs}, 'kernels': res}
# HBM bytes one train step moves: every libfsmg kernel of a pass (FETCH_SIZE + WRITE_SIZE, the two passes of one family) over its steps
for fam, n in step_counts.items():
    f, w = totals.get('FETCH_SIZE_' + fam), totals.get('WRITE_SIZE_' + fam)
    if f and w:
        doc.setdefault('step_hbm_bytes_by_family', {})[fam] = {'fetch': f / n, 'write': w / n, 'total': (f + w) / n, 'steps': n}
if 'f32cell' in doc.get('step_hbm_bytes_by_family', {}):
    doc['step_hbm_bytes_measured'] = doc['step_hbm_bytes_by_family']['f32cell']['total']
    doc['step_hbm_bytes_note'] = 'serial order (FSMG_XCD_OVERLAP=0), every k_* dispatch of the pass incl. handle creation, over its train steps'
json.dump(doc, open(out_path, 'w'), indent=1, sort_keys=True)
print('%-64s %7s %10s %12s %12s %9s' % ('kernel', 'calls', 'us', 'FETCH_KB', 'WRITE_KB', 'mfma'))
for name, e in sorted(res.items(), key=lambda kv: -kv[1]['mean_duration_us'] * kv[1]['launches']):
    print('%-64s %7d %10.1f %12.0f %12.0f %9s' % (name[:64], e['launches'], e['mean_duration_us'], e.get('FETCH_SIZE_KB', float('nan')),
                                                 e.get('WRITE_SIZE_KB', float('nan')), ('%.3f' % e['mfma_busy_frac']) if 'mfma_busy_frac' in e else '-'))
