#!/usr/bin/env python3
"""Race hunt, stage 2: on a FRESH handle, (a) first forward_backward: states + gradients; (b) then apply_update:
parameters; (c) second train step: parameters.  Which stage first differs from repetition 0?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = dict(bench.CFG_B); N, K, Q = 5, 5, 4; B = N * (K + Q)
eps = bench.synthetic_episodes(2, N, K, Q, cfg['max_len'], cfg['input_size'], 5)
ref = None; bad = 0
def diff(a, b):
    return [k for k in a if a[k].tobytes() != b[k].tobytes()]
for r in range(reps):
    m = FsmgModel(cfg); m.init_params(3)
    d = m.debug_dims(); T, Hp = d['T'], d['Hp']
    m.forward_backward(*eps[0])
    st = {'h': m.debug_read('h0', (T + 1) * B * Hp), 'dz': m.debug_read('gates0', T * B * 4 * Hp), 'dh': m.debug_read('dh', T * B * Hp)}
    g1 = {k: m.get_grad(k) for k in m.param_shapes}
    m.apply_update(1.0)
    p1 = m.get_params()
    if os.environ.get('HUNT_SPLIT'):
        m.forward_backward(*eps[1])
        g2 = {k: m.get_grad(k) for k in m.param_shapes}
        m.apply_update(1.0)
    else:
        m.train_step(*eps[1])
        g2 = {}
    p2 = m.get_params()
    cur = (st, g1, p1, g2, p2)
    if ref is None: ref = cur
    else:
        names = ['states after pass 1', 'gradients of pass 1', 'parameters after update 1', 'gradients of pass 2', 'parameters after step 2']
        msgs = ['%s: %s' % (n, diff(c, q)) for n, c, q in zip(names, cur, ref) if diff(c, q)]
        if msgs:
            bad += 1
            print('rep %d DIFFERS | ' % r + ' | '.join(msgs))
            for c, q in zip(cur[1:], ref[1:]):
                for k in diff(c, q):
                    x, y = c[k], q[k]
                    ne = np.argwhere(x != y)
                    print('    %s %s: %d elements differ, rows %d..%d cols %d..%d, max |d| %.3g (|ref| max %.3g)' % (
                        k, x.shape, len(ne), ne[:, 0].min(), ne[:, 0].max(), ne[:, -1].min(), ne[:, -1].max(), float(np.abs(x - y).max()), float(np.abs(y).max())))
            a = cur[0]['h'].reshape(T + 1, B, Hp); b = ref[0]['h'].reshape(T + 1, B, Hp)
            ne = a != b
            if ne.any():
                ts = np.where(ne.reshape(T + 1, -1).any(axis=1))[0]; t = ts[0]
                rows = np.where(ne[t].any(axis=1))[0]; cols = np.where(ne[t].any(axis=0))[0]
                print('    h: time indices %d..%d differ (%d of them); at %d: rows %s, cols %s%s, max |d| %.3g, sample %r vs %r' % (
                    ts[0], ts[-1], len(ts), t, rows[:10].tolist(), cols[:16].tolist(), '...' if len(cols) > 16 else '',
                    float(np.abs(a[t] - b[t]).max()), a[t][ne[t]][:3].tolist(), b[t][ne[t]][:3].tolist()))
    st_ = m.stats()
    if st_['timeouts'] or st_['steps_skipped_timeout']: print('rep %d: stats %r' % (r, st_))
    m.close()
print('reps', reps, 'mismatching', bad)
