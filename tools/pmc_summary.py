#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 *_counter_collection.csv."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r'fsmg::\(anonymous namespace\)::', '', r['Kernel_Name'])[:60]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
counters = sorted({c for k in agg.values() for c in k})
print('%-60s %6s ' % ('kernel', 'calls') + ' '.join('%14s' % c[-14:] for c in counters))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', [0]))):
    n = max(len(x) for x in v.values())
    print('%-60s %6d ' % (k, n) + ' '.join('%14.0f' % (sum(v.get(c, [0])) / max(len(v.get(c, [1])), 1)) for c in counters))
