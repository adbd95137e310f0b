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


def _torch_gate_perm(H):
    """column blocks (i, j, f, o) of BasicLSTMCell -> torch.nn.LSTM's row blocks (i, f, g, o)  """
    i, j, f, o = (np.arange(H) + k * H for k in range(4))
    return np.concatenate([i, f, j, o])


@pytest.mark.parametrize('L', [1, 2])
def test_forward_matches_torch_nn_lstm(L):
    """An implementation that shares NO formula with lstm_oracle.py: torch.nn.LSTM (ATen's fused cell) +
    F.cross_entropy, fed the reference-layout weights through the SURVEY A.5 mapping
    (weight_ih = reorder(K[:in]).T, weight_hh = reorder(K[in:]).T, bias_ih = reorder(b + forget_bias on f), bias_hh = 0).
    Pins gate order, the run-time forget bias, the output row order b*T+t and the mean-over-all-tokens loss."""
    import torch.nn.functional as F
    cfg = small_config(n_layers=L, input_size=37, embedding_size=8, hidden_size=16, max_len=10)
    d = O.model_dims(cfg)
    H, T = d['H'], d['T']
    params = O.glorot_init(cfg, 11, np.float64)
    for l in range(L):       # non-zero biases so that a wrong forget-bias placement cannot hide
        params['bias_%d' % l] = np.random.RandomState(l).uniform(-0.5, 0.5, 4 * H)
    sup, qry = _episode(cfg, N=2, K=2, Q=1, seed=5)
    X, Y = O.train_xy(sup, qry, cfg['input_size'])
    want, cache = O.forward(params, X, Y, cfg)

    perm = _torch_gate_perm(H)
    lstm = torch.nn.LSTM(d['E'], H, num_layers=L, batch_first=True, dtype=torch.float64)
    with torch.no_grad():
        for l in range(L):
            K, b = params['kernel_%d' % l], params['bias_%d' % l]
            n_in = d['E'] if l == 0 else H
            fb = np.zeros(4 * H); fb[2 * H:3 * H] = O.FORGET_BIAS
            getattr(lstm, 'weight_ih_l%d' % l).copy_(torch.from_numpy(K[:n_in][:, perm].T.copy()))
            getattr(lstm, 'weight_hh_l%d' % l).copy_(torch.from_numpy(K[n_in:][:, perm].T.copy()))
            getattr(lstm, 'bias_ih_l%d' % l).copy_(torch.from_numpy((b + fb)[perm]))
            getattr(lstm, 'bias_hh_l%d' % l).zero_()
        emb = torch.from_numpy(params['embedding'])[torch.from_numpy(X.astype(np.int64))]      # [B,T,E]
        out, (hn, cn) = lstm(emb)                                                              # zero initial state
        logits = out.reshape(-1, H) @ torch.from_numpy(params['softmax_w']) + torch.from_numpy(params['softmax_b'])
        got = F.cross_entropy(logits, torch.from_numpy(Y.reshape(-1).astype(np.int64)), reduction='mean')
    assert abs(float(got) - float(want)) <= 1e-10 * abs(float(want))
    np.testing.assert_allclose(out.numpy(), np.transpose(cache['layers'][-1]['hs'][1:], (1, 0, 2)), rtol=0, atol=1e-12)
    np.testing.assert_allclose(cn[-1].numpy(), cache['layers'][-1]['cs'][T], rtol=0, atol=1e-12)


def test_adam_and_lr_decay_match_torch_optim_adam():
    """torch.optim.Adam shares no code with apply_update.  TF1's Adam (reference lstm_baseline.py:82) is
    theta -= lr_t * m / (sqrt(v) + eps), lr_t = lr_s sqrt(1-b2^t)/(1-b1^t): torch's update with eps_torch =
    eps / sqrt(1-b2^t) (set per step below -> agreement to rounding).  The decayed learning rate lr * 0.5 ** (s / n_decay)
    (continuous, s = global_step BEFORE the update; reference lstm_baseline.py:77-81) is fed to torch per step;
    n_decay = 3 makes a wrong base / exponent / off-by-one visible at once (factor 0.79 per step)."""
    cfg = small_config(lr=5e-3, n_decay=3, max_grad_norm=1e9)
    rng = np.random.RandomState(0)
    w0 = rng.uniform(-1, 1, 50)
    params = {'w': w0.copy()}
    opt = O.new_opt_state(params)
    tw = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    topt = torch.optim.Adam([tw], lr=1.0, betas=(O.BETA1, O.BETA2), eps=0.0)
    for s in range(6):
        g = rng.uniform(0.5, 1.5, 50) * rng.choice([-1, 1], 50)
        for grp in topt.param_groups:
            grp['lr'] = 5e-3 * 0.5 ** (s / 3.0)
            grp['eps'] = O.ADAM_EPS / np.sqrt(1.0 - O.BETA2 ** (s + 1))     # TF's epsilon placement in torch's terms
        tw.grad = torch.from_numpy(g.copy())
        topt.step()
        O.apply_update(params, {'w': g}, {'embedding_slices_sq': 0.0}, opt, cfg, 'dense')
        np.testing.assert_allclose(params['w'], tw.detach().numpy(), rtol=0, atol=1e-14)
    assert opt['step'] == 6
    assert np.abs(params['w'] - w0).max() > 5e-3          # the parameters moved by far more than the tolerance


@pytest.mark.parametrize('inner_steps', [1, 2])
def test_maml_first_order_gradients_match_torch_autograd(inner_steps):
    """cfg-E (BASELINE.json configs[4]; semantics in DESIGN.md): theta' by clipped SGD on the support rows, the outer
    gradient is d L_query / d theta' -- checked against autograd with the inner steps detached, then one outer update."""
    cfg = small_config(n_layers=2, max_grad_norm=0.05)         # small clip: the inner clip branch is active
    params = O.glorot_init(cfg, 13, np.float64)
    sup, qry = _episode(cfg, N=2, K=2, Q=2, seed=9)
    loss, grads, aux = O.maml_query_grads(params, sup, qry, cfg, inner_steps=inner_steps, inner_lr=0.2)
    ref = TorchRef(cfg, params, dtype=torch.float64)
    rloss, rg, rsq = ref.maml_query_grads(sup, qry, inner_steps, 0.2)
    assert abs(loss - rloss) <= 1e-12 * abs(rloss)
    for k in grads:
        np.testing.assert_allclose(grads[k], rg[k].numpy(), rtol=1e-9, atol=1e-13, err_msg=k)
    assert abs(aux['embedding_slices_sq'] - rsq) <= 1e-10 * rsq
    # the adaptation helps on the support set it was computed from, and theta itself is untouched
    before = {k: v.copy() for k, v in params.items()}
    fast, sup_losses = O.maml_adapt(params, sup, cfg, inner_steps=3, inner_lr=0.2)
    assert sup_losses[0] > sup_losses[1] > sup_losses[2]
    for k in params:
        np.testing.assert_array_equal(params[k], before[k])
    # one outer step == apply_update with those gradients
    opt, opt2 = O.new_opt_state(params), O.new_opt_state(params)
    p2 = {k: v.copy() for k, v in params.items()}
    got = O.maml_step(params, opt, sup, qry, cfg, inner_steps=inner_steps, inner_lr=0.2)
    O.apply_update(p2, grads, aux, opt2, cfg)
    assert got == loss and opt['step'] == 1
    for k in params:
        np.testing.assert_array_equal(params[k], p2[k])
    assert O.maml_eval(before, sup, qry, cfg, inner_steps=inner_steps, inner_lr=0.2) == pytest.approx(loss, rel=1e-12)
