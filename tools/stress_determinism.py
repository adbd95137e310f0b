#!/usr/bin/env python3
"""Race hunt: full-size cfg-B (two-stream schedule) and cfg-C (graph replay) steps repeated on fresh handles must be
bit-identical in losses and in every gradient / parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
only = os.environ.get('STRESS_ONLY')
for name, (cfg, N, K, Q) in (('cfg-B', (dict(bench.CFG_B), 5, 5, 4)), ('cfg-C', bench.OTHER['cfg-C']), ('cfg-D', bench.OTHER['cfg-D'])):
    if only and name != only: continue
    eps = bench.synthetic_episodes(3, N, K, Q, cfg['max_len'], cfg['input_size'], 5)
    ref = None
    bad = 0
    for r in range(reps):
        m = FsmgModel(cfg); m.init_params(3)
        losses = [m.train_step(s, q) for s, q in eps]
        m.forward_backward(*eps[0])
        sig = (losses, {k: m.get_grad(k).tobytes() for k in m.param_shapes}, float(m.eval_step(eps[1][1])))
        if ref is None: ref = sig
        elif sig[0] != ref[0] or sig[2] != ref[2] or any(sig[1][k] != ref[1][k] for k in ref[1]):
            bad += 1
            print(name, 'rep', r, 'DIFFERS', sig[0], ref[0], [k for k in ref[1] if sig[1][k] != ref[1][k]])
        m.close()
    print(name, 'reps', reps, 'mismatches', bad, 'losses', ref[0])
