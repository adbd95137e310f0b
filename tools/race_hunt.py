#!/usr/bin/env python3
"""Where does a non-reproducible cfg-B forward/backward first differ?  Fresh handle per repetition (same seed), one
forward_backward, then h / c / logits / dH are compared with the first repetition's, time step by time step."""
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
for r in range(reps):
    if os.environ.get('HUNT_POISON'):      # a handle with OTHER weights uses (and frees) the same device addresses first
        p = FsmgModel(cfg); p.init_params(100 + r); p.forward_backward(*eps[1]); p.synchronize(); p.close()
    m = FsmgModel(cfg); m.init_params(3)
    if os.environ.get('HUNT_WARM'):
        saved = m.get_params()
        m.train_step(*eps[0]); m.train_step(*eps[1])          # previous passes leave their data in every buffer ...
        for k, v in saved.items(): m.set_param(k, v)          # ... but the compared pass starts from identical parameters
    m.debug_set('inplace_dlogits', 0)       # the comparison reads the logits
    m.forward_backward(*eps[0])
    d = m.debug_dims(); T, Hp, V1p = d['T'], d['Hp'], d['V1p']
    cur = {'h': m.debug_read('h0', (T + 1) * B * Hp).reshape(T + 1, B, Hp), 'c': m.debug_read('c0', (T + 1) * B * Hp).reshape(T + 1, B, Hp),
           'logits': m.debug_read('logits', T * B * V1p).reshape(T, B, V1p), 'dh': m.debug_read('dh', T * B * Hp).reshape(T, B, Hp),
           'dz': m.debug_read('gates0', T * B * 4 * Hp).reshape(T, B, 4 * Hp)}
    if ref is None: ref = cur
    else:
        msgs = []
        for k in cur:
            ne = cur[k] != ref[k]
            if ne.any():
                ts = np.where(ne.reshape(ne.shape[0], -1).any(axis=1))[0]
                t = ts[0]; rows = np.where(ne[t].any(axis=1))[0]; cols = np.where(ne[t].any(axis=0))[0]
                msgs.append('%s: %d time indices differ, first %d last %d (at first: rows %s cols %s%s, %d elements, max |d| %.3g)' % (
                    k, len(ts), t, ts[-1], rows[:8].tolist(), cols[:12].tolist(), '...' if len(cols) > 12 else '', int(ne[t].sum()), float(np.abs(cur[k][t] - ref[k][t]).max())))
        if msgs:
            bad += 1
            print('rep %d DIFFERS\n   ' % r + '\n   '.join(msgs))
    m.close()
print('reps', reps, 'mismatching', bad)
