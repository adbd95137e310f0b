"""The numpy restatement vs an independent torch-autograd implementation, and
vs closed-form anchors.  (CPU only.)"""
import numpy as np
import pytest
import torch

from conftest import small_config
from oracle import lstm_oracle as O
from oracle.torch_ref import TorchRef


def _episode(cfg, N=2, K=2, Q=1, seed=0):
    rng = np.random.RandomState(seed)
    V, T = cfg['input_size'], cfg['max_len']
    sup = rng.randint(0, V, size=(N, K, T)).astype(np.int32)
    qry = rng.randint(0, V, size=(N, Q, T)).astype(np.int32)
    sup[0, 0, T // 2:] = 0          # zero padding is NOT masked (base_loader.py:59-61)
    return sup, qry


@pytest.mark.parametrize('L', [1, 2])
def test_forward_and_grads_match_torch_autograd(L):
    cfg = small_config(n_layers=L)
    params = O.glorot_init(cfg, 7, np.float64)
    sup, qry = _episode(cfg)
    X, Y = O.train_xy(sup, qry, cfg['input_size'])
    loss, cache = O.forward(params, X, Y, cfg)
    grads, aux = O.backward(params, cache, cfg)
    ref = TorchRef(cfg, params, dtype=torch.float64)
    rloss, rg, rsq = ref.grads_xy(X, Y)
    assert abs(loss - rloss) <= 1e-12 * abs(rloss)
    for k in grads:
        np.testing.assert_allclose(grads[k], rg[k].numpy(), rtol=1e-9, atol=1e-13, err_msg=k)
    assert abs(aux['embedding_slices_sq'] - rsq) <= 1e-10 * rsq
    # the START row is hit B times at t=0 -> the slices norm differs from the dense norm (Q7)
    dense_sq = float((grads['embedding'] ** 2).sum())
    assert abs(aux['embedding_slices_sq'] - dense_sq) > 1e-3 * dense_sq


@pytest.mark.parametrize('mode', ['tf1_slices', 'dense'])
def test_ten_update_trajectory_matches_torch(mode):
    cfg = small_config(max_grad_norm=0.05)      # small clip so the clip branch is active
    params = O.glorot_init(cfg, 3, np.float64)
    ref = TorchRef(cfg, params, dtype=torch.float64, clip_norm_mode=mode)
    opt = O.new_opt_state(params)
    for s in range(10):
        sup, qry = _episode(cfg, seed=s)
        a = O.train_step(params, opt, sup, qry, cfg, clip_norm_mode=mode)
        b = ref.train(sup, qry)
        assert abs(a - b) <= 1e-10 * abs(b), (s, a, b)
    for k, v in ref.numpy_params().items():
        np.testing.assert_allclose(params[k], v, rtol=1e-8, atol=1e-12, err_msg=k)
    assert opt['step'] == 10


def test_untrained_nll_is_about_log_vocab():
    cfg = small_config(input_size=999, hidden_size=8, embedding_size=4)
    params = O.glorot_init(cfg, 1, np.float64)
    _, qry = _episode(cfg)
    nll = O.eval_step(params, qry, cfg)
    assert abs(nll - np.log(1000)) < 0.05       # ln(V1) anchor


def test_fp32_oracle_close_to_fp64():
    cfg = small_config()
    p64 = O.glorot_init(cfg, 5, np.float64)
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    sup, qry = _episode(cfg)
    a, b = O.eval_step(p64, qry, cfg), O.eval_step(p32, qry, cfg)
    assert abs(a - b) <= 1e-5 * abs(a)


def test_gate_order_and_forget_bias():
    """One cell step by hand: gate order i,j,f,o; forget bias added at run time."""
    cfg = small_config(input_size=3, embedding_size=2, hidden_size=1, max_len=1)
    params = O.glorot_init(cfg, 0, np.float64)
    params['kernel_0'][:] = 0.0
    params['bias_0'][:] = np.array([0.3, -0.2, 0.1, 0.7])       # i, j, f, o
    X = np.array([[3]]); Y = np.array([[1]])
    _, cache = O.forward(params, X, Y, cfg)
    sig = lambda x: 1 / (1 + np.exp(-x))
    c = sig(0.3) * np.tanh(-0.2)                                 # c_prev = 0 so f is unused here
    h = np.tanh(c) * sig(0.7)
    assert abs(cache['layers'][0]['hs'][1][0, 0] - h) < 1e-15
    assert abs(cache['layers'][0]['gates'][0, 2][0, 0] - sig(0.1 + 1.0)) < 1e-15


def test_sample_is_greedy_and_deterministic():
    cfg = small_config()
    params = O.glorot_init(cfg, 2, np.float64)
    s1, s2 = O.sample(params, 12, cfg), O.sample(params, 12, cfg)
    assert s1 == s2 and len(s1) == 12
    assert all(0 <= w <= cfg['input_size'] for w in s1)


def test_learning_rate_schedule_and_adam_first_step():
    cfg = small_config(lr=5e-3, n_decay=10000)
    assert O.learning_rate(cfg, 0) == 5e-3
    assert abs(O.learning_rate(cfg, 10000) - 2.5e-3) < 1e-18
    # first Adam step moves every touched weight by ~alpha (m/sqrt(v) = +-1), eps outside
    params = {'w': np.array([1.0])}
    opt = O.new_opt_state(params)
    O.apply_update(params, {'w': np.array([1e-3])}, {'embedding_slices_sq': 0.0}, opt, cfg, 'dense')
    alpha = 5e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    m, v = 0.1 * 1e-3, 0.001 * 1e-6
    assert abs(params['w'][0] - (1.0 - alpha * m / (np.sqrt(v) + 1e-8))) < 1e-15
