#!/usr/bin/env python3
"""Freeze the known-answer vectors of the LSTM-baseline arithmetic (fixture G5).

    python tests/golden/make_golden_g5.py        -> tests/golden/g5_lstm.npz

The reference's own regression test (src/train/test_seed.py:45-65) pins ten consecutive train
losses of the TensorFlow graph; TensorFlow is absent here, so what CAN be frozen is the fp64
restatement (oracle/lstm_oracle.py) at the moment of authoring: a small 2-layer model's initial
parameters (fp32 values, so the HIP path starts from the same bits), ten episodes, and for them
  * per-step h and c of both layers, the per-row cross entropy and the loss of the first train
    batch at the initial parameters, every gradient tensor of that batch and its global norm
    (both clip modes),
  * the query-set NLL (LSTMBaseline.eval) at the initial parameters,
  * ten consecutive train losses, the learning rate of every step (n_decay = 7: the decay moves),
    and the parameters + Adam moments after the tenth update, for both clip-norm modes (Q7).
Every GPU parity test recomputes the oracle live; this file is what catches a simultaneous wrong
edit of oracle and kernel (tests/test_golden_g5.py: the live oracle must still reproduce it, and
the HIP path is compared with the FILE).  Data only: no reference code is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import lstm_oracle as O      # noqa: E402

CFG = dict(name='lstm_baseline', seed=1234, input_size=41, max_len=10, embedding_size=8, hidden_size=12,
           n_layers=2, lr=5e-3, max_grad_norm=0.25, n_decay=7)
N, K, Q, STEPS = 2, 2, 2, 10


def build():
    p32 = {k: v.astype(np.float32) for k, v in O.glorot_init(CFG, 20261001, np.float64).items()}
    episodes = O.synthetic_episodes(STEPS, N, K, Q, CFG['max_len'], CFG['input_size'], seed=606, realistic=True)
    out = dict(cfg_keys=np.array(sorted(CFG)), cfg_vals=np.array([str(CFG[k]) for k in sorted(CFG)]),
               shape=np.array([N, K, Q, STEPS]),
               support=np.stack([s for s, _ in episodes]).astype(np.int32),
               query=np.stack([q for _, q in episodes]).astype(np.int32))
    for k, v in p32.items():
        out['init/' + k] = v
    params = {k: v.astype(np.float64) for k, v in p32.items()}
    sup, qry = episodes[0]
    X, Y = O.train_xy(sup, qry, CFG['input_size'])
    loss, cache = O.forward(params, X, Y, CFG)
    grads, aux = O.backward(params, cache, CFG)
    out['first/loss'] = np.float64(loss)
    out['first/ce'] = cache['ce']
    out['first/lse'] = cache['lse']
    for l, lay in enumerate(cache['layers']):
        out['first/h%d' % l] = lay['hs']
        out['first/c%d' % l] = lay['cs']
    for k, g in grads.items():
        out['first/grad/' + k] = g
    out['first/slices_sq'] = np.float64(aux['embedding_slices_sq'])
    for mode in ('tf1_slices', 'dense'):
        out['first/gnorm/' + mode] = np.float64(O.global_norm(grads, aux, mode))
    out['first/eval_nll'] = np.float64(O.eval_step(params, qry, CFG))
    for mode in ('tf1_slices', 'dense'):
        p = {k: v.astype(np.float64) for k, v in p32.items()}
        opt = O.new_opt_state(p)
        losses, lrs = [], []
        for s, (sup, qry) in enumerate(episodes):
            lrs.append(O.learning_rate(CFG, opt['step']))
            losses.append(O.train_step(p, opt, sup, qry, CFG, clip_norm_mode=mode))
        out['traj/%s/losses' % mode] = np.array(losses)
        out['traj/%s/lr' % mode] = np.array(lrs)
        for k in p:
            out['traj/%s/param/%s' % (mode, k)] = p[k]
            if mode == 'tf1_slices':              # (the moments once: < 200 KB of fixture)
                out['traj/%s/m/%s' % (mode, k)] = opt['m'][k]
                out['traj/%s/v/%s' % (mode, k)] = opt['v'][k]
    return out


if __name__ == '__main__':
    data = build()
    path = os.path.join(HERE, 'g5_lstm.npz')
    np.savez_compressed(path, **data)
    print('wrote %s (%d arrays, %d bytes)' % (path, len(data), os.path.getsize(path)))
