#!/usr/bin/env python3
"""Per-kernel-class time of the validation path (fsmg_eval_batch, 16 episodes per pass) at cfg-B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
cfg = dict(bench.CFG_B)
m = FsmgModel(cfg); m.init_params(1)
q = np.stack([e[1] for e in bench.synthetic_episodes(16, 5, 5, 4, cfg['max_len'], cfg['input_size'], 3)])
m.eval_batch(q)
for mode in ('default', 'timed'):
    if mode == 'timed':
        m.timing_select(None); m.timing_enable(True); m.timing_reset()
    t0 = time.perf_counter()
    for _ in range(5): m.eval_batch(q)
    dt = (time.perf_counter() - t0) / 5
    print('%s: %.3f ms per 16-episode pass -> %.0f eval episodes/s' % (mode, dt * 1e3, 16 / dt))
for c in ('gemm_zx', 'lstm_fwd', 'gemm_logits', 'ce'):
    ms, n = m.timing_read(c)
    print('  %-12s %.3f ms per pass' % (c, ms / 5))
