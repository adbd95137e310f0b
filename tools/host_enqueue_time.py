#!/usr/bin/env python3
"""Is the eager two-stream schedule host-bound?  Host time to ENQUEUE one cfg-B train step (no sync) vs the GPU
time per step, for FSMG_OVERLAP=1 (eager launches) and =0 (one hipGraph launch per phase)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import torch
import bench
from fsmg.binding import FsmgModel
cfg = dict(bench.CFG_B)
m = FsmgModel(cfg); m.init_params(1)
eps = bench.synthetic_episodes(8, 5, 5, 4, cfg['max_len'], cfg['input_size'], 1234)
for sup, qry in eps: m.train_step(sup, qry)
m.synchronize()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n):
    sup, qry = eps[i % len(eps)]
    m.train_step(sup, qry, want_loss=False)
t1 = time.perf_counter()
torch.cuda.synchronize(); m.synchronize()
t2 = time.perf_counter()
print('FSMG_OVERLAP=%s: host enqueue %.3f ms/step, enqueue+drain %.3f ms/step' % (os.environ.get('FSMG_OVERLAP', 'default'), (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
