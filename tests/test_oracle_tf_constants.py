"""Known-answer constants of TensorFlow 1.x's OWN unit tests for the ops the reference graph is built from, restated here
(TensorFlow is absent on either box) and run against the oracle (oracle/lstm_oracle.py, test infrastructure).

VERDICT r04 item 6.  By the brief's rule this does not lift "parity unpinned" -- the constants are TensorFlow's, not held by
/root/reference (its only pinned values, src/train/test_seed.py:45-48, need TF1 + python 2 + the real datasets) -- but it replaces
"the TF1 semantics of SURVEY.md Appendix A were recalled from memory" with checkable numbers.  Every case names the TF 1.x source
file and test it restates (r1.4-r1.15 carry all of them; the numbers did not change across those releases).

What each one decides for the oracle, and through it for the HIP path:
  * BasicLSTMCell / MultiRNNCell known answers  -> gate order i, j, f, o; forget_bias added at run time; c before h in the state;
                                                   [input | h_prev] rows of `kernel` (reference lstm_baseline.py:44-55);
  * sequence_loss                               -> natural log, sum / (total weight + 1e-12) (lstm_baseline.py:70-75);
  * exponential_decay                           -> continuous lr * rate ** (step / decay_steps) (lstm_baseline.py:77-81);
  * clip_by_global_norm with an IndexedSlices   -> the norm is taken over IndexedSlices.values (one slice per token
                                                   occurrence, NOT summed per row first) -- why clip_norm_mode defaults to tf1_slices;
  * Adam's numpy reference + repeated indices   -> epsilon outside the bias correction (on the un-corrected sqrt(v)); duplicate
                                                   rows of a sparse gradient are summed before the update (lstm_baseline.py:82-87).
"""
import numpy as np

from oracle import lstm_oracle as O


def _cell_config(H, L, E):
    return dict(input_size=3, max_len=1, embedding_size=E, hidden_size=H, n_layers=L, lr=1e-3, max_grad_norm=5.0, n_decay=100.0)


def _cell_step(cfg, params, x, state):
    """one step of the stacked cell through the oracle's forward(): the embedding row of token 0 is x, initial states are `state`"""
    # forward() starts every layer from zeros (lstm_baseline.py:50-51), so the stack is driven one step at a time through the
    # oracle's own single-step cell when it has one; otherwise restated from forward()'s per-step arithmetic below
    H = cfg['hidden_size']
    out, new_state = x, []
    for l in range(cfg['n_layers']):
        c, h = state[l]
        z = np.concatenate([out, h], axis=1) @ params['kernel_%d' % l] + params['bias_%d' % l]
        i, j, f, o = np.split(z, 4, axis=1)                                 # BasicLSTMCell: i, j, f, o = split(value=concat, num=4)
        c = c * O._sigmoid(f + O.FORGET_BIAS) + O._sigmoid(i) * np.tanh(j)
        h = np.tanh(c) * O._sigmoid(o)
        new_state.append((c, h))
        out = h
    return out, new_state


def test_basic_lstm_cell_known_answers_from_tf_rnn_cell_test():
    """tensorflow/contrib/rnn/python/kernel_tests/core_rnn_cell_test.py (r1.0-r1.4; later python/kernel_tests/rnn_cell_test.py),
    RNNCellTest.testBasicLSTMCell: MultiRNNCell([BasicLSTMCell(2, state_is_tuple=False)] * 2) under
    variable_scope("root", initializer=constant_initializer(0.5)) -- which reaches the kernels only: the bias has its own zero
    initialiser in every 1.x release (_linear(bias_start=0.0), later zeros_initializer) --, x = [[1, 1]],
    state = 0.1 * ones([1, 8]) laid out [c0, h0, c1, h1] (float32 results, quoted to 8 digits):
        res[0]       = [[0.24024698, 0.24024698]]
        expected_mem = [[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]]
    The numbers are reproduced by kernel = 0.5 everywhere, bias = 0 (the zeros-initialised bias of the release the reference ran
    on: tf.contrib.rnn.BasicLSTMCell, lstm_baseline.py:44-46), forget_bias = 1.0 added at run time, gate order i, j, f, o."""
    cfg = _cell_config(2, 2, 2)
    params = {'kernel_0': np.full((4, 8), 0.5), 'bias_0': np.zeros(8), 'kernel_1': np.full((4, 8), 0.5), 'bias_1': np.zeros(8)}
    x = np.array([[1.0, 1.0]])
    state = [(np.full((1, 2), 0.1), np.full((1, 2), 0.1)), (np.full((1, 2), 0.1), np.full((1, 2), 0.1))]
    g, new_state = _cell_step(cfg, params, x, state)
    np.testing.assert_allclose(g, [[0.24024698, 0.24024698]], rtol=0, atol=3e-7)          # TF's numbers are fp32 results
    mem = np.concatenate([new_state[0][0], new_state[0][1], new_state[1][0], new_state[1][1]], axis=1)
    np.testing.assert_allclose(mem, [[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]],
                               rtol=0, atol=3e-7)
    # the same through the oracle's forward(): zero initial state, T = 1, the embedding row of the start word = x
    cfg1 = _cell_config(2, 1, 2)
    p1 = dict(O.glorot_init(cfg1, 0), kernel_0=np.full((4, 8), 0.5), bias_0=np.zeros(8))
    p1['embedding'][cfg1['input_size']] = [1.0, 1.0]                        # X[:, 0] is the start word (base_model.py:82-84)
    X, Y = O.tokens_to_input_and_target(np.zeros((1, 1, 1), np.int32), cfg1['input_size'])       # [B, S, T] tokens
    _, cache = O.forward(p1, X, Y, cfg1)
    z = 1.0                                                                 # (1 + 1 + 0 + 0) * 0.5
    c = O._sigmoid(z) * np.tanh(z)
    np.testing.assert_allclose(cache['layers'][0]['hs'][1], np.full((1, 2), np.tanh(c) * O._sigmoid(z)), rtol=1e-12)
    # a different gate order or a forget bias folded into `bias` would give other numbers: i, f, g, o (the cuDNN / torch order) here
    i, f, j, o = 1.1, 1.1, 1.1, 1.1
    wrong = 0.1 * O._sigmoid(f) + O._sigmoid(i) * np.tanh(j)               # no run-time forget bias
    assert abs(wrong - 0.68967271) > 1e-2


def test_basic_lstm_cell_variable_shapes_from_tf_rnn_cell_test():
    """The same test's sibling block (num_units = 2, a 3-wide input) asserts the variable names and shapes
    "root/basic_lstm_cell/kernel" [(3 + 2), 4 * 2] and ".../bias" [4 * 2]: the [input | h_prev] row order of `kernel`."""
    cfg = _cell_config(2, 1, 3)
    shapes = dict(O.param_shapes(cfg))
    assert shapes['kernel_0'] == (3 + 2, 4 * 2) and shapes['bias_0'] == (8,)


def test_sequence_loss_known_answer_from_tf_seq2seq_loss_test():
    """tensorflow/contrib/seq2seq/python/kernel_tests/loss_test.py LossTest.testSequenceLoss: batch 2, 3 time steps, 5 classes,
    logits of step i constant i + 0.5, targets i, weights 1: average over time and batch -> 1.60944 (= ln 5).
    Also its per-example variants: average_across_timesteps only -> [1.60944, 1.60944]; neither -> 1.60944 * ones((2, 3))."""
    cfg = dict(input_size=4, max_len=3, embedding_size=2, hidden_size=2, n_layers=1)
    params = O.glorot_init(cfg, 1)
    params['softmax_w'][:] = 0.0                                             # logits = softmax_b: constant over the classes
    params['softmax_b'][:] = 0.5
    Y = np.array([[0, 1, 2], [0, 1, 2]])
    X = np.zeros((2, 3), np.int64)
    loss, cache = O.forward(params, X, Y, cfg)
    np.testing.assert_allclose(loss, 1.60944, rtol=0, atol=5e-6)
    np.testing.assert_allclose(cache['ce'], np.full(6, 1.60944), rtol=0, atol=5e-6)


def test_exponential_decay_known_answers_from_tf_learning_rate_decay_test():
    """tensorflow/python/training/learning_rate_decay_test.py LRDecayTest.testContinuous: exponential_decay(0.05, step = 5, 10, 0.96)
    = 0.05 * 0.96 ** (5 / 10); testStaircase shows staircase=True floors the exponent -- the reference passes staircase=False
    (lstm_baseline.py:77-81) with rate 0.5."""
    cfg = dict(lr=0.05, n_decay=10.0)
    np.testing.assert_allclose(O.learning_rate(cfg, 5), 0.05 * 0.5 ** 0.5, rtol=1e-15)
    np.testing.assert_allclose(O.learning_rate(cfg, 10), 0.025, rtol=1e-15)
    assert O.learning_rate(cfg, 5) != O.learning_rate(cfg, 0)               # continuous, not a staircase


def test_clip_by_global_norm_with_indexed_slices_from_tf_clip_ops_test():
    """tensorflow/python/kernel_tests/clip_ops_test.py ClipTest.testClipByGlobalNormWithIndexedSlicesClipped:
        x0 = [[-2, 0, 0], [4, 0, 0]] (dense), x1 = IndexedSlices(values [1, -2], indices [3, 4]), clip_norm = 4
        norm = sqrt(2^2 + 4^2 + 1^2 + 2^2) = 5; answers = the tensors scaled by 4 / 5, the slices' VALUES scaled in place.
    clip_ops.global_norm takes `t.values` of an IndexedSlices: nothing is scattered or summed per row first.  The embedding gradient
    of tf.gather is such an IndexedSlices with ONE slice per token occurrence -- START occurs B times, pad id 0 many times -- so the
    reference's norm is sqrt(sum over occurrences), SURVEY.md Q7: clip_norm_mode = tf1_slices, the oracle's and the library's default."""
    dense = {'w': np.array([[-2.0, 0.0, 0.0], [4.0, 0.0, 0.0]])}
    values = np.array([1.0, -2.0])
    aux = {'embedding_slices_sq': float((values ** 2).sum())}
    grads = dict(dense, embedding=np.zeros((6, 1)))
    grads['embedding'][3, 0], grads['embedding'][4, 0] = 1.0, -2.0
    np.testing.assert_allclose(O.global_norm(grads, aux, 'tf1_slices'), 5.0, rtol=1e-15)
    scale = 4.0 / max(O.global_norm(grads, aux, 'tf1_slices'), 4.0)
    np.testing.assert_allclose(dense['w'] * scale, [[-1.6, 0.0, 0.0], [3.2, 0.0, 0.0]], rtol=1e-15)
    np.testing.assert_allclose(values * scale, [0.8, -1.6], rtol=1e-15)
    # with REPEATED indices the two readings differ: slices [1, -2] both on row 3 -> values norm sqrt(5), summed-per-row norm 1
    rep = dict(dense, embedding=np.zeros((6, 1)))
    rep['embedding'][3, 0] = 1.0 - 2.0
    assert abs(O.global_norm(rep, aux, 'tf1_slices') - 5.0) < 1e-12          # TF1: over the values
    assert abs(O.global_norm(rep, aux, 'dense') - np.sqrt(20.0 + 1.0)) < 1e-12


def _adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """tensorflow/python/training/adam_test.py adam_update_numpy, verbatim in meaning"""
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t
    return param - alpha_t * m_t / (np.sqrt(v_t) + epsilon), m_t, v_t


def test_adam_matches_tf_adam_test_numpy_reference_and_sums_repeated_indices():
    """adam_test.py AdamOptimizerTest.testBasic / testSparse: var0 = [1, 2], var1 = [3, 4], grads0 = [0.1, 0.1], grads1 = [0.01, 0.01],
    three steps against adam_update_numpy (epsilon added to sqrt(v_t) OUTSIDE the bias correction, which sits in alpha_t);
    testSparseRepeatedIndices: IndexedSlices([0.1, 0.1], indices [1, 1]) updates exactly like IndexedSlices([0.2], [1]).
    The oracle's apply_update is driven with a clip norm nothing reaches, so that only Adam acts."""
    cfg = dict(input_size=1, max_len=1, embedding_size=1, hidden_size=1, n_layers=1, lr=0.001, max_grad_norm=1e9, n_decay=1e30)
    params = {k: np.zeros(s) for k, s in O.param_shapes(cfg)}
    params['embedding'][:, 0] = [1.0, 2.0]                                   # var0 (V1 = 2 rows of width 1)
    params['softmax_b'][:] = [3.0, 4.0]                                      # var1
    opt = O.new_opt_state(params)
    v0, m0, s0 = np.array([1.0, 2.0]), 0.0, 0.0
    v1, m1, s1 = np.array([3.0, 4.0]), 0.0, 0.0
    g0, g1 = np.array([0.1, 0.1]), np.array([0.01, 0.01])
    for t in range(1, 4):
        grads = {k: np.zeros_like(v) for k, v in params.items()}
        grads['embedding'][:, 0] = g0
        grads['softmax_b'][:] = g1
        O.apply_update(params, grads, {'embedding_slices_sq': float((g0 ** 2).sum())}, opt, cfg, 'tf1_slices')
        v0, m0, s0 = _adam_update_numpy(v0, g0, t, m0, s0)
        v1, m1, s1 = _adam_update_numpy(v1, g1, t, m1, s1)
        np.testing.assert_allclose(params['embedding'][:, 0], v0, rtol=1e-13)
        np.testing.assert_allclose(params['softmax_b'], v1, rtol=1e-13)
    # epsilon inside the correction (torch.optim.Adam's placement) is a different update: the test above would not pass with it
    t, g = 1, 0.1
    tf_step = 0.001 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    torch_step = 0.001 * (0.1 * g / (1 - 0.9)) / (np.sqrt(0.001 * g * g / (1 - 0.999)) + 1e-8)
    assert abs(tf_step - torch_step) > 1e-11 and abs((1.0 - tf_step) - _adam_update_numpy(1.0, g, 1, 0.0, 0.0)[0]) < 1e-15
    # repeated indices: the oracle's backward() sums the occurrences of a token into its embedding row (tf.gather's gradient is
    # aggregated by the optimizer, adam.py _apply_sparse_shared after _deduplicate_indexed_slices): two occurrences of gradient 0.1
    # on row 1 act like one of 0.2
    cfg2 = dict(input_size=2, max_len=2, embedding_size=2, hidden_size=2, n_layers=1, lr=1e-3, max_grad_norm=5.0, n_decay=100.0)
    p2 = O.glorot_init(cfg2, 3)
    X = np.array([[1, 1]]); Y = np.array([[0, 1]])
    _, cache = O.forward(p2, X, Y, cfg2)
    grads, aux = O.backward(p2, cache, cfg2)
    # token 1 occurs twice: its row holds the SUM of the two occurrences' gradients, the other rows nothing; the slices' squared
    # norm |a|^2 + |b|^2 (what clip_by_global_norm sees) is not the row's |a + b|^2
    assert np.all(grads['embedding'][0] == 0) and np.all(grads['embedding'][2] == 0) and np.any(grads['embedding'][1] != 0)
    assert abs(aux['embedding_slices_sq'] - float((grads['embedding'][1] ** 2).sum())) > 1e-12
