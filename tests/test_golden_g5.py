"""Fixture G5: frozen known-answer vectors of the LSTM-baseline arithmetic -- tests/golden/g5_lstm.npz, written by
tests/golden/make_golden_g5.py from the fp64 restatement at authoring time (the reference's own regression pins ten consecutive train
losses of the TF graph: src/train/test_seed.py:45-65; TensorFlow is absent, so the restatement's numbers are what can be frozen).

CPU: the LIVE oracle must still reproduce the file (an edit of oracle/lstm_oracle.py that changes any number fails here).
GPU: the HIP path through the C-ABI is compared with the FILE, not with a live oracle run -- so a simultaneous wrong edit of the
oracle and a kernel cannot pass both."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import GOLDEN

G5 = os.path.join(GOLDEN, 'g5_lstm.npz')


def _load():
    d = np.load(G5)
    cfg = {}
    for k, v in zip(d['cfg_keys'], d['cfg_vals']):
        k, v = str(k), str(v)
        cfg[k] = v if k == 'name' else (int(v) if v.lstrip('-').isdigit() else float(v))
    return d, cfg


def test_live_oracle_reproduces_the_frozen_vectors():
    spec = importlib.util.spec_from_file_location('make_golden_g5', os.path.join(GOLDEN, 'make_golden_g5.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    live = mod.build()
    d, cfg = _load()
    assert cfg == mod.CFG
    assert sorted(live) == sorted(d.files)
    for k in d.files:
        if d[k].dtype.kind in 'iUS':
            np.testing.assert_array_equal(np.asarray(live[k]), d[k], err_msg=k)
        else:       # libm differences between numpy builds: a few ulps of exp / tanh, far below every parity tolerance
            np.testing.assert_allclose(np.asarray(live[k]), d[k], rtol=1e-9, atol=1e-13, err_msg=k)


def test_fixture_is_small_and_self_consistent():
    d, cfg = _load()
    assert os.path.getsize(G5) < 200 * 1024
    N, K, Q, steps = (int(x) for x in d['shape'])
    assert d['support'].shape == (steps, N, K, cfg['max_len']) and d['query'].shape == (steps, N, Q, cfg['max_len'])
    assert abs(float(d['first/loss']) - float(d['first/ce'].sum()) / (d['first/ce'].size + 1e-12)) < 1e-12
    for mode in ('tf1_slices', 'dense'):
        assert float(d['traj/%s/losses' % mode][0]) == float(d['first/loss'])      # the first train loss is the loss at the initial parameters
        assert abs(d['traj/%s/lr' % mode][7] - 0.5 * cfg['lr']) < 1e-15             # n_decay = 7: the rate has halved by step 7
    assert float(d['first/gnorm/tf1_slices']) != float(d['first/gnorm/dense'])       # Q7: the two norms differ
    assert float(d['first/gnorm/dense']) > cfg['max_grad_norm']                      # ... and the clip is active


@pytest.mark.gpu
def test_hip_forward_backward_matches_the_frozen_vectors():
    from gpu_utils import new_model, read_states, rel_max, time_major
    d, cfg = _load()
    N, K, Q, _ = (int(x) for x in d['shape'])
    B, T = N * (K + Q), cfg['max_len']
    init = {k[5:]: d[k] for k in d.files if k.startswith('init/')}
    model = new_model(cfg, params=init, max_sequences=B)
    sup, qry = d['support'][0], d['query'][0]
    nll = model.eval_step(qry)
    assert abs(nll - float(d['first/eval_nll'])) <= 1e-4 * float(d['first/eval_nll'])
    model.forward_backward(sup, qry)
    tail = model.debug_read('tail', 16)
    assert abs(tail[1] - float(d['first/loss'])) <= 1e-4 * float(d['first/loss'])
    assert abs(tail[0] - float(d['first/slices_sq'])) <= 1e-4 * float(d['first/slices_sq'])
    for l in range(cfg['n_layers']):
        hs, cs, _ = read_states(model, cfg, l, B)
        assert rel_max(hs, d['first/h%d' % l]) < 2e-5, 'h layer %d' % l
        assert rel_max(cs, d['first/c%d' % l]) < 2e-5, 'c layer %d' % l
    assert rel_max(model.debug_read('ce', B * T), time_major(d['first/ce'], B, T)) < 1e-5
    assert rel_max(model.debug_read('lse', B * T), time_major(d['first/lse'], B, T)) < 1e-5
    for k in d.files:
        if k.startswith('first/grad/'):
            assert rel_max(model.get_grad(k[11:]), d[k]) < 2e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tf1_slices', 'dense'])
def test_hip_ten_update_trajectory_matches_the_frozen_vectors(mode):
    """the reference's regression shape (test_seed.py:45-65: ten consecutive train losses from fixed parameters and episodes) against
    the frozen numbers: each loss within 1e-4 relative, then the parameters and Adam moments after the tenth update; the learning rate
    decays by 2^(1/7) per step (n_decay = 7) and the clip is active at every step"""
    from gpu_utils import new_model, rel_max
    d, cfg = _load()
    N, K, Q, steps = (int(x) for x in d['shape'])
    init = {k[5:]: d[k] for k in d.files if k.startswith('init/')}
    model = new_model(cfg, params=init, clip_norm_mode=mode, max_sequences=N * (K + Q))
    want = d['traj/%s/losses' % mode]
    for s in range(steps):
        got = model.train_step(d['support'][s], d['query'][s])
        assert abs(got - want[s]) <= 1e-4 * abs(want[s]), (s, got, want[s])
        if s == 0:        # the pre-clip global norm of the first update, in this mode's definition (Q7)
            assert abs(float(model.debug_read('gnorm', 1)[0]) - float(d['first/gnorm/' + mode])) <= 1e-4 * float(d['first/gnorm/' + mode])
    assert model.step == steps
    for name in init:
        assert rel_max(model.get_param(name), d['traj/%s/param/%s' % (mode, name)]) < 5e-4, name
        if mode == 'tf1_slices':
            m, v = model.get_opt_state(name)
            assert rel_max(m, d['traj/%s/m/%s' % (mode, name)]) < 2e-3, name
            assert rel_max(v, d['traj/%s/v/%s' % (mode, name)]) < 4e-3, name
