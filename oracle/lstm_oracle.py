"""numpy restatement of the reference LSTM-baseline graph (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines (under /root/reference) it restates.
The arithmetic itself lives in TensorFlow 1.x (absent, unpinned); its semantics
are restated from SURVEY.md Appendix A and cross-checked against torch autograd
in tests/test_oracle.py.  Parity vs TF itself: unpinned (see oracle/__init__.py).

dtype is a parameter: float64 is the "truth" the HIP path is compared with
(NLL within 1e-4 relative), float32 shows what plain fp32 arithmetic gives.
"""
import numpy as np

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults (lstm_baseline.py:82)
FORGET_BIAS = 1.0                            # BasicLSTMCell(forget_bias=1.) (lstm_baseline.py:45-46)


# ----------------------------------------------------------------------------- config
def model_dims(config):
    """Sizes read by the plugin (src/models/lstm_baseline.py:21-29)."""
    return dict(
        V1=int(config['input_size']) + 1,      # +1 start word (:22)
        start=int(config['input_size']),       # (:21)
        T=int(config['max_len']),
        E=int(config['embedding_size']),
        H=int(config['hidden_size']),
        L=int(config['n_layers']),
    )


def param_shapes(config):
    """Trainable variables in creation order (lstm_baseline.py:39-40,44-49,60-62;
    tf_model.py:99-104).  BasicLSTMCell kernel is [(in+H), 4H], bias [4H]."""
    d = model_dims(config)
    shapes = [('embedding', (d['V1'], d['E']))]
    for l in range(d['L']):
        fan_in = d['E'] if l == 0 else d['H']
        shapes.append(('kernel_%d' % l, (fan_in + d['H'], 4 * d['H'])))
        shapes.append(('bias_%d' % l, (4 * d['H'],)))
    shapes.append(('softmax_w', (d['H'], d['V1'])))
    shapes.append(('softmax_b', (d['V1'],)))
    return shapes


def glorot_init(config, seed, dtype=np.float64):
    """TF1 default initializer (none is passed: lstm_baseline.py:39-40,60-62):
    Glorot-uniform for every get_variable incl. the 1-D softmax_b (fan_in =
    fan_out = V1); LSTM bias zeros.  TF's own RNG stream is not reproducible, so
    only the distribution is restated."""
    rng = np.random.RandomState(seed)
    params = {}
    for name, shape in param_shapes(config):
        if name.startswith('bias_'):
            params[name] = np.zeros(shape, dtype)
            continue
        fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else shape
        limit = np.sqrt(6.0 / (fan_in + fan_out))
        params[name] = rng.uniform(-limit, limit, size=shape).astype(dtype)
    return params


# ----------------------------------------------------------------------------- tokens
def flatten_first_two_dims(tokens):
    """[B,S,N] -> [B*S,N]  (src/models/base_model.py:57-60)."""
    tokens = np.asarray(tokens)
    return tokens.reshape(tokens.shape[0] * tokens.shape[1], tokens.shape[2])


def tokens_to_input_and_target(tokens, start_word):
    """src/models/base_model.py:63-86 with a start word: target = the songs,
    input = the songs shifted right by one with start_word in column 0."""
    flat = flatten_first_two_dims(tokens)
    y = flat.copy()
    x = np.empty_like(flat)
    x[:, 0] = start_word
    x[:, 1:] = flat[:, :-1]
    return x, y


def train_xy(support, query, start_word):
    """LSTMBaseline.train feed (lstm_baseline.py:91-96): support rows first."""
    xs, ys = tokens_to_input_and_target(support, start_word)
    xq, yq = tokens_to_input_and_target(query, start_word)
    return np.concatenate([xs, xq]), np.concatenate([ys, yq])


def eval_xy(query, start_word):
    """LSTMBaseline.eval feed (lstm_baseline.py:117-118): query only."""
    return tokens_to_input_and_target(query, start_word)


# ----------------------------------------------------------------------------- forward
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def forward(params, X, Y, config, keep=True):
    """Embedding -> L x BasicLSTMCell static unroll -> xw_plus_b -> sequence_loss
    (lstm_baseline.py:39-75).  Zero initial state, no
    masking of padding, all-ones weights.  Returns (loss, cache)."""
    d = model_dims(config)
    H, L, T = d['H'], d['L'], d['T']
    dtype = params['embedding'].dtype
    X = np.asarray(X)
    Y = np.asarray(Y)
    B = X.shape[0]
    assert X.shape == (B, T) and Y.shape == (B, T)

    layer_in = params['embedding'][X]                 # [B,T,E]  embedding_lookup (:41)
    layers = []
    for l in range(L):
        K, b = params['kernel_%d' % l], params['bias_%d' % l]
        n_in = layer_in.shape[2]
        Kx, Kh = K[:n_in], K[n_in:]
        zx = layer_in.reshape(B * T, n_in).dot(Kx).reshape(B, T, 4 * H) + b
        h = np.zeros((B, H), dtype)
        c = np.zeros((B, H), dtype)
        hs = np.zeros((T + 1, B, H), dtype)           # hs[t+1] = h_t, hs[0] = 0
        cs = np.zeros((T + 1, B, H), dtype)
        gates = np.zeros((T, 4, B, H), dtype)         # activated i, j, f, o
        for t in range(T):
            z = zx[:, t] + h.dot(Kh)
            i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]   # gate order i,j,f,o
            si, tj, sf, so = _sigmoid(i), np.tanh(j), _sigmoid(f + FORGET_BIAS), _sigmoid(o)
            c = c * sf + si * tj
            h = np.tanh(c) * so
            hs[t + 1], cs[t + 1] = h, c
            gates[t, 0], gates[t, 1], gates[t, 2], gates[t, 3] = si, tj, sf, so
        layers.append(dict(x=layer_in, hs=hs, cs=cs, gates=gates))
        layer_in = np.transpose(hs[1:], (1, 0, 2))     # [B,T,H]

    out = layer_in.reshape(B * T, H)                   # row = b*T + t (:57-58)
    logits = out.dot(params['softmax_w']) + params['softmax_b']
    m = logits.max(axis=1)
    lse = m + np.log(np.exp(logits - m[:, None]).sum(axis=1))
    yflat = Y.reshape(B * T)
    ce = lse - logits[np.arange(B * T), yflat]
    loss = ce.sum() / (B * T + 1e-12)                  # sequence_loss averaging (:70-75)
    cache = None
    if keep:
        cache = dict(X=X, Y=Y, layers=layers, out=out, logits=logits, lse=lse, ce=ce, B=B)
    return dtype.type(loss), cache


# ----------------------------------------------------------------------------- backward
def backward(params, cache, config):
    """tf.gradients of the loss w.r.t. every trainable var (lstm_baseline.py:83-84;
    SURVEY.md A.3).  Returns (grads, aux): grads['embedding'] is the dense,
    per-row-summed gradient; aux['embedding_slices_sq'] is the squared norm of
    the un-deduplicated IndexedSlices values (one slice per token occurrence),
    which is what TF1's clip_by_global_norm sees."""
    d = model_dims(config)
    H, L, T, V1 = d['H'], d['L'], d['T'], d['V1']
    B = cache['B']
    dtype = params['embedding'].dtype
    n = B * T
    grads = {}

    p = np.exp(cache['logits'] - cache['lse'][:, None])
    p[np.arange(n), cache['Y'].reshape(n)] -= 1.0
    dlogits = p / dtype.type(n + 1e-12)
    grads['softmax_w'] = cache['out'].T.dot(dlogits)
    grads['softmax_b'] = dlogits.sum(axis=0)
    dtop = dlogits.dot(params['softmax_w'].T).reshape(B, T, H)      # [B,T,H]

    for l in reversed(range(L)):
        lay = cache['layers'][l]
        K = params['kernel_%d' % l]
        n_in = lay['x'].shape[2]
        Kh = K[n_in:]
        dz_all = np.zeros((B, T, 4 * H), dtype)
        dh_rec = np.zeros((B, H), dtype)
        dc = np.zeros((B, H), dtype)
        for t in reversed(range(T)):
            si, tj, sf, so = lay['gates'][t]
            c_t, c_prev = lay['cs'][t + 1], lay['cs'][t]
            tc = np.tanh(c_t)
            dh = dtop[:, t] + dh_rec
            do = dh * tc * so * (1.0 - so)
            dc = dc + dh * so * (1.0 - tc * tc)
            di = dc * tj * si * (1.0 - si)
            dj = dc * si * (1.0 - tj * tj)
            df = dc * c_prev * sf * (1.0 - sf)
            dz = np.concatenate([di, dj, df, do], axis=1)
            dz_all[:, t] = dz
            dh_rec = dz.dot(Kh.T)
            dc = dc * sf
        dzf = dz_all.reshape(n, 4 * H)
        hprev = np.transpose(lay['hs'][:-1], (1, 0, 2)).reshape(n, H)
        xin = lay['x'].reshape(n, n_in)
        grads['kernel_%d' % l] = np.concatenate([xin.T.dot(dzf), hprev.T.dot(dzf)], axis=0)
        grads['bias_%d' % l] = dzf.sum(axis=0)
        dtop = dzf.dot(K[:n_in].T).reshape(B, T, n_in)

    dx = dtop.reshape(n, -1)                            # one slice per token occurrence
    demb = np.zeros_like(params['embedding'])
    np.add.at(demb, cache['X'].reshape(n), dx)
    grads['embedding'] = demb
    aux = dict(embedding_slices_sq=float((dx.astype(np.float64) ** 2).sum()))
    return grads, aux


# ----------------------------------------------------------------------------- update
def new_opt_state(params):
    return dict(step=0,
                m={k: np.zeros_like(v) for k, v in params.items()},
                v={k: np.zeros_like(v) for k, v in params.items()})


def global_norm(grads, aux, clip_norm_mode='tf1_slices'):
    """tf.clip_by_global_norm's norm (lstm_baseline.py:83-85).  'tf1_slices':
    the embedding term is the IndexedSlices.values norm (Q7); 'dense': the
    per-row-summed dense gradient's norm."""
    sq = 0.0
    for k, g in grads.items():
        if k == 'embedding' and clip_norm_mode == 'tf1_slices':
            sq += aux['embedding_slices_sq']
        else:
            sq += float((g.astype(np.float64) ** 2).sum())
    return np.sqrt(sq)


def learning_rate(config, step):
    """tf.train.exponential_decay(lr, global_step, n_decay, 0.5, staircase=False)
    (lstm_baseline.py:77-81)."""
    return float(config['lr']) * 0.5 ** (step / float(config['n_decay']))


def apply_update(params, grads, aux, opt, config, clip_norm_mode='tf1_slices'):
    """clip_by_global_norm then TF-style Adam (epsilon outside the bias
    correction), then global_step += 1 (lstm_baseline.py:77-87).
    Mutates params/opt in place; returns the pre-clip global norm."""
    clip = float(config['max_grad_norm'])
    gnorm = global_norm(grads, aux, clip_norm_mode)
    scale = clip / max(gnorm, clip)
    s = opt['step']
    t = s + 1
    lr_s = learning_rate(config, s)
    alpha = lr_s * np.sqrt(1.0 - BETA2 ** t) / (1.0 - BETA1 ** t)
    for k in params:
        dt = params[k].dtype.type
        g = grads[k] * dt(scale)
        opt['m'][k] = dt(BETA1) * opt['m'][k] + dt(1.0 - BETA1) * g
        opt['v'][k] = dt(BETA2) * opt['v'][k] + dt(1.0 - BETA2) * g * g
        params[k] = params[k] - dt(alpha) * opt['m'][k] / (np.sqrt(opt['v'][k]) + dt(ADAM_EPS))
    opt['step'] = t
    return gnorm


# ----------------------------------------------------------------------------- plugin-level calls
def train_step(params, opt, support, query, config, clip_norm_mode='tf1_slices'):
    """LSTMBaseline.train (lstm_baseline.py:89-113): returns the loss computed
    with the PRE-update parameters, then applies one Adam step."""
    d = model_dims(config)
    X, Y = train_xy(support, query, d['start'])
    loss, cache = forward(params, X, Y, config)
    grads, aux = backward(params, cache, config)
    apply_update(params, grads, aux, opt, config, clip_norm_mode)
    return float(loss)


def eval_step(params, query, config):
    """LSTMBaseline.eval (lstm_baseline.py:115-133): query-only mean NLL."""
    d = model_dims(config)
    X, Y = eval_xy(query, d['start'])
    loss, _ = forward(params, X, Y, config, keep=False)
    return float(loss)


def sample(params, num, config):
    """LSTMBaseline.sample (lstm_baseline.py:135-156): greedy argmax decode from
    the start word and a zero state; the support set is ignored (Q8)."""
    d = model_dims(config)
    H, L = d['H'], d['L']
    dtype = params['embedding'].dtype
    hs = [np.zeros(H, dtype) for _ in range(L)]
    cs = [np.zeros(H, dtype) for _ in range(L)]
    word = d['start']
    out = []
    for _ in range(num):
        x = params['embedding'][word]
        for l in range(L):
            z = np.concatenate([x, hs[l]]).dot(params['kernel_%d' % l]) + params['bias_%d' % l]
            i, j, f, o = z[:H], z[H:2 * H], z[2 * H:3 * H], z[3 * H:]
            cs[l] = cs[l] * _sigmoid(f + FORGET_BIAS) + _sigmoid(i) * np.tanh(j)
            hs[l] = np.tanh(cs[l]) * _sigmoid(o)
            x = hs[l]
        logits = x.dot(params['softmax_w']) + params['softmax_b']
        word = int(np.argmax(logits))                   # argmax of softmax == argmax of logits
        out.append(word)
    return out


# ----------------------------------------------------------------------------- cfg-E: MAML-style inner / outer loop
# BASELINE.json configs[4].  The reference has no code for it (READING_LIST.md:5-7 names MAML as the direction); the
# semantics are specified in DESIGN.md section "cfg-E" and restated here.  First-order MAML on the LSTM baseline:
#   theta'  <- theta;  repeat inner_steps times on the SUPPORT rows:
#                theta' <- theta' - inner_lr * clip_by_global_norm(grad L_support(theta'), max_grad_norm)
#   loss_q  =  L_query(theta')                      (the value train() returns; pre-outer-update)
#   g       =  grad_{theta'} L_query(theta')        (first-order: no derivative through the inner steps)
#   theta   <- clip + TF-Adam(theta, mean over ranks of g);  global_step += 1      (the same update as LSTMBaseline.train)
def maml_adapt(params, support, config, inner_steps=1, inner_lr=0.1, clip_norm_mode='tf1_slices'):
    """-> (theta', [support loss before each inner step]); params is not modified."""
    d = model_dims(config)
    fast = {k: v.copy() for k, v in params.items()}
    X, Y = tokens_to_input_and_target(support, d['start'])
    losses = []
    clip = float(config['max_grad_norm'])
    for _ in range(int(inner_steps)):
        loss, cache = forward(fast, X, Y, config)
        grads, aux = backward(fast, cache, config)
        scale = clip / max(global_norm(grads, aux, clip_norm_mode), clip)
        for k in fast:
            dt = fast[k].dtype.type
            fast[k] = fast[k] - dt(inner_lr) * (grads[k] * dt(scale))
        losses.append(float(loss))
    return fast, losses


def maml_query_grads(params, support, query, config, inner_steps=1, inner_lr=0.1, clip_norm_mode='tf1_slices'):
    """-> (query loss at theta', first-order outer gradients, aux)"""
    d = model_dims(config)
    fast, _ = maml_adapt(params, support, config, inner_steps, inner_lr, clip_norm_mode)
    X, Y = tokens_to_input_and_target(query, d['start'])
    loss, cache = forward(fast, X, Y, config)
    grads, aux = backward(fast, cache, config)
    return float(loss), grads, aux


def maml_step(params, opt, support, query, config, inner_steps=1, inner_lr=0.1, clip_norm_mode='tf1_slices'):
    """One outer step; mutates params / opt; returns the query loss at the adapted parameters."""
    loss, grads, aux = maml_query_grads(params, support, query, config, inner_steps, inner_lr, clip_norm_mode)
    apply_update(params, grads, aux, opt, config, clip_norm_mode)
    return loss


def maml_eval(params, support, query, config, inner_steps=1, inner_lr=0.1, clip_norm_mode='tf1_slices'):
    """Few-shot evaluation: adapt on the support set, mean NLL of the query set at theta'; no state change."""
    fast, _ = maml_adapt(params, support, config, inner_steps, inner_lr, clip_norm_mode)
    return eval_step(fast, query, config)


# ----------------------------------------------------------------------------- synthetic workloads
def synthetic_episodes(n_episodes, N, K, Q, T, vocab, seed=1234, realistic=False):
    """cfg-B style inputs: ids i.i.d. uniform on [0, vocab) from RandomState(seed).
    realistic=True: Zipf(1.1) ids, song length ~ U[T/4, T], zero padded (exercises
    duplicate embedding rows and the unmasked-padding path)."""
    rng = np.random.RandomState(seed)
    eps = []
    for _ in range(n_episodes):
        if not realistic:
            sup = rng.randint(0, vocab, size=(N, K, T)).astype(np.int32)
            qry = rng.randint(0, vocab, size=(N, Q, T)).astype(np.int32)
        else:
            def songs(n):
                a = np.minimum(rng.zipf(1.1, size=(N, n, T)) - 1, vocab - 1).astype(np.int32)
                lens = rng.randint(max(1, T // 4), T + 1, size=(N, n))
                a[np.arange(T)[None, None, :] >= lens[:, :, None]] = 0
                return a
            sup, qry = songs(K), songs(Q)
        eps.append((sup, qry))
    return eps
