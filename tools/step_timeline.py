#!/usr/bin/env python3
"""Compressed timeline of one train step from a rocprofv3 rocpd trace (kernels by queue)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_token_prep' in r[0]]
seg = rows[idx[which]:idx[which + 1]]; t0 = seg[0][1]
def short(n):
    n = re.sub(r'fsmg::\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n); return n.replace('void ', '')
out = []
for name, s, e, q in seg:
    n = short(name)
    if out and out[-1][0] == n and out[-1][4] == q: out[-1][2] = (e - t0) / 1e3; out[-1][3] += 1
    else: out.append([n, (s - t0) / 1e3, (e - t0) / 1e3, 1, q])
for o in out: print('%-30s q%-2s %8.1f -> %8.1f us  x%-3d %s' % (o[0][:30], o[4], o[1], o[2], o[3], ('%.1f us/launch' % ((o[2] - o[1]) / o[3])) if o[3] > 1 else ''))
print('step span %.1f us' % ((seg[-1][2] - t0) / 1e3))
# period = start of the next step's first kernel - start of this one's: the gap between two steps is period - span (copies, event
# records, launch boundaries); mean over the steps around the one printed
per = [(rows[idx[i + 1]][1] - rows[idx[i]][1]) / 1e3 for i in range(max(which - 8, 0), min(which + 8, len(idx) - 1))]
if per: print('step period %.1f us (median of %d consecutive steps; min %.1f max %.1f)' % (sorted(per)[len(per) // 2], len(per), min(per), max(per)))
