#!/usr/bin/env python3
"""A long training run (cfg-B dims; LONG_RUN_CONFIG=cfg-C | ref-default for the others) on the default path (XCD-partitioned order, fused softmax): the loss of a small pool of structured
synthetic episodes must go down, nothing may be skipped, and the tallies that would say a silent fall-back happened must stay zero.
   python tools/long_run.py [steps]      -> loss every 250 steps, fsmg_stats at the end, episodes/s of the whole run"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
from oracle import lstm_oracle as O
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
which = os.environ.get('LONG_RUN_CONFIG', 'cfg-B')        # cfg-B | cfg-C | ref-default
cfg = dict(bench.CFG_B if which == 'cfg-B' else bench.OTHER[which][0], lr=1e-3)
eps = O.synthetic_episodes(32, 5, 5, 4, cfg['max_len'], cfg['input_size'], seed=11, realistic=True)
m = FsmgModel(cfg); m.init_params(3)
first = float(np.mean([m.eval_step(q) for _, q in eps[:8]]))
t0 = time.perf_counter()
window = []
for s in range(steps):
    sup, qry = eps[s % len(eps)]
    want = (s % 250) >= 242
    l = m.train_step(sup, qry, want_loss=want)
    if want: window.append(l)
    if s % 250 == 249:
        print('step %5d  mean loss of the last 8 steps %.4f' % (s + 1, float(np.mean(window)))); window = []
m.synchronize()
dt = time.perf_counter() - t0
fused = bool(m.debug_read('fused_softmax', 2)[1])      # (of the last TRAIN pass: the evaluations below reset it)
last = float(np.mean([m.eval_step(q) for _, q in eps[:8]]))
st = m.stats()
print('eval NLL of 8 pool episodes: %.4f before, %.4f after %d steps' % (first, last, steps))
print('episodes/s over the whole run (numpy token arrays staged from pageable host memory on every call, a loss read back on 8 of 250 steps): %.1f' % (steps / dt))
print('global_step', m.step, {k: st[k] for k in ('timeouts', 'steps_skipped_timeout', 'steps_skipped_token_range', 'xov_selfcheck_mismatches', 'softmax_range_rows', 'persistent_path', 'xcd_launches')},
      'fused softmax taken:', fused)
ok = m.step == steps and st['timeouts'] == 0 and st['steps_skipped_timeout'] == 0 and st['xov_selfcheck_mismatches'] == 0 and st['softmax_range_rows'] == 0 and fused and last < first - 1.0
print('LONG_RUN_OK' if ok else 'LONG_RUN_FAILED')
sys.exit(0 if ok else 1)
