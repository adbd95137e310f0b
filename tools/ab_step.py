#!/usr/bin/env python3
"""Same-box, same-process A/B of train-step variants: one handle per variant (the FSMG_* environment is read when a handle is
created), timed regions interleaved A B C A B C ... so that clock / thermal drift hits every variant alike.

  python tools/ab_step.py [--config cfg-B] [--steps 30] [--rounds 5] name:ENV=V,ENV=V name2: ...
  e.g.  python tools/ab_step.py base:FSMG_UPD_SPLIT=0,FSMG_TAIL_ASIDE=0,FSMG_INPLACE_DLOGITS=0 all: aside:FSMG_UPD_SPLIT=0,FSMG_INPLACE_DLOGITS=0
Prints one line per variant: median / min / max ms per step over the rounds, episodes/s, the guard (steps advanced, time-outs)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np          # noqa: E402
import bench                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg-B')
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('variants', nargs='+')
args = ap.parse_args()

import torch                # noqa: E402
from fsmg.dist import EpisodeParallel          # noqa: E402
from models.lstm_baseline import LSTMBaseline  # noqa: E402
from models.maml_lstm import MAMLLSTM          # noqa: E402

base, N, K, Q = (dict(bench.CFG_B), 5, 5, 4) if args.config == 'cfg-B' else bench.OTHER[args.config]
cfg = dict(base, device=0)
maml = (cfg['inner_steps'], cfg['inner_lr']) if args.config == 'cfg-E' else None
pool = bench.synthetic_episodes(64, N, K, Q, cfg['max_len'], cfg['input_size'], seed=99)
d_sup = torch.from_numpy(np.stack([s for s, _ in pool])).cuda()
d_qry = torch.from_numpy(np.stack([q for _, q in pool])).cuda()
ss, qs = d_sup[0].numel() * 4, d_qry[0].numel() * 4
kw = dict(maml=maml) if maml else {}
runs = []
for v in args.variants:
    name, _, envs = v.partition(':')
    env = dict(e.split('=', 1) for e in envs.split(',') if e)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    m = (MAMLLSTM if maml else LSTMBaseline)(dict(cfg, max_sequences=N * (K + Q)))
    m.recover_or_init('')
    for k, old in saved.items():
        if old is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = old
    par = EpisodeParallel(m)
    runs.append(dict(name=name, env=env, m=m, par=par, ms=[], i=0))


def go(r, n):
    for _ in range(n):
        e = r['i'] % len(pool)
        r['par'].train_step(d_sup.data_ptr() + e * ss, d_qry.data_ptr() + e * qs, want_loss=False, shape=(N, K, Q), **kw)
        r['i'] += 1


for r in runs:
    go(r, 12)
    torch.cuda.synchronize()          # (handles must never run side by side: persistent kernels of two handles time each other out)
    r['m'].engine.synchronize()
for rnd in range(args.rounds):
    for r in (runs if rnd % 2 == 0 else runs[::-1]):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(r, args.steps)
        torch.cuda.synchronize()
        r['ms'].append(1e3 * (time.perf_counter() - t0) / args.steps)
ref = None
for r in runs:
    ms = sorted(r['ms'])
    med = ms[len(ms) // 2]
    ref = ref or med
    st = r['m'].engine.stats()
    print('   rounds: ' + ' '.join('%.3f' % x for x in r['ms']))
    print('%-14s %.4f ms/step (min %.4f max %.4f)  %7.1f episodes/s  %+5.1f %% vs %s   steps %d timeouts %d skipped %d selfcheck %d  %s'
          % (r['name'], med, ms[0], ms[-1], 1e3 / med, 100.0 * (ref / med - 1.0), runs[0]['name'], r['m'].engine.step, st['timeouts'],
             st['steps_skipped_timeout'], st['xov_selfcheck_mismatches'], ' '.join('%s=%s' % kv for kv in sorted(r['env'].items()))))
