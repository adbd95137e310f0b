"""torch-CPU autograd implementation of the same graph (TEST INFRASTRUCTURE ONLY).

Two jobs:
  1. independent cross-check of oracle/lstm_oracle.py (forward NLL, every
     gradient, the clip + TF-style Adam update) — tests/test_oracle.py;
  2. the `cpu_baseline` leg of bench.py ("port": TensorFlow cannot run on either
     box, so the reference's CPU path is timed as this restatement of the
     identical op graph on the host cores; closest stand-in for TF-Eigen/MKL).

Follows /root/reference/src/models/lstm_baseline.py:38-87 (graph) and :89-133
(feeds); TF1 semantics per SURVEY.md Appendix A.  It deliberately does NOT use
torch.nn.LSTM / torch.optim.Adam: the cell is written with the reference's gate
order (i, j, f, o) and run-time forget bias, and the optimizer with TF's epsilon
placement.
"""
import numpy as np
import torch

from . import lstm_oracle as O


class TorchRef(object):
    def __init__(self, config, params, dtype=torch.float32, threads=None, clip_norm_mode='tf1_slices'):
        if threads:
            torch.set_num_threads(int(threads))
        self.config = dict(config)
        self.d = O.model_dims(config)
        self.dtype = dtype
        self.clip_norm_mode = clip_norm_mode
        self.names = [n for n, _ in O.param_shapes(config)]
        self.p = {k: torch.tensor(np.asarray(params[k]), dtype=dtype, requires_grad=True) for k in self.names}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.step = 0

    # -- graph --------------------------------------------------------------
    def _nll(self, X, Y, emb_rows=None):
        d, p = self.d, self.p
        H, L, T = d['H'], d['L'], d['T']
        X = torch.as_tensor(np.asarray(X), dtype=torch.long)
        Y = torch.as_tensor(np.asarray(Y), dtype=torch.long)
        B = X.shape[0]
        x = p['embedding'][X] if emb_rows is None else emb_rows      # [B,T,E]
        for l in range(L):
            K, b = p['kernel_%d' % l], p['bias_%d' % l]
            n_in = x.shape[2]
            zx = (x.reshape(B * T, n_in) @ K[:n_in]).reshape(B, T, 4 * H) + b
            Kh = K[n_in:]
            h = torch.zeros(B, H, dtype=self.dtype)
            c = torch.zeros(B, H, dtype=self.dtype)
            outs = []
            for t in range(T):
                z = zx[:, t] + h @ Kh
                i, j, f, o = z.split(H, dim=1)
                c = c * torch.sigmoid(f + O.FORGET_BIAS) + torch.sigmoid(i) * torch.tanh(j)
                h = torch.tanh(c) * torch.sigmoid(o)
                outs.append(h)
            x = torch.stack(outs, dim=1)
        logits = x.reshape(B * T, H) @ p['softmax_w'] + p['softmax_b']
        ce = torch.logsumexp(logits, dim=1) - logits.gather(1, Y.reshape(B * T, 1)).squeeze(1)
        return ce.sum() / (B * T + 1e-12)

    def eval_xy(self, X, Y):
        with torch.no_grad():
            return float(self._nll(X, Y))

    def grads_xy(self, X, Y):
        """loss, dense grads, and the IndexedSlices squared norm for the embedding."""
        for v in self.p.values():
            v.grad = None
        Xl = torch.as_tensor(np.asarray(X), dtype=torch.long)
        rows = self.p['embedding'][Xl]
        rows.retain_grad()
        loss = self._nll(X, Y, emb_rows=rows)
        loss.backward()
        slices_sq = float((rows.grad.double() ** 2).sum())
        return float(loss.detach()), {k: v.grad for k, v in self.p.items()}, slices_sq

    # -- update ---------------------------------------------------------------
    def train_xy(self, X, Y):
        loss, g, slices_sq = self.grads_xy(X, Y)
        cfg = self.config
        sq = 0.0
        for k in self.names:
            sq += slices_sq if (k == 'embedding' and self.clip_norm_mode == 'tf1_slices') \
                else float((g[k].double() ** 2).sum())
        gnorm = np.sqrt(sq)
        clip = float(cfg['max_grad_norm'])
        scale = clip / max(gnorm, clip)
        t = self.step + 1
        alpha = O.learning_rate(cfg, self.step) * np.sqrt(1.0 - O.BETA2 ** t) / (1.0 - O.BETA1 ** t)
        with torch.no_grad():
            for k in self.names:
                gk = g[k] * scale
                self.m[k].mul_(O.BETA1).add_(gk, alpha=1.0 - O.BETA1)
                self.v[k].mul_(O.BETA2).addcmul_(gk, gk, value=1.0 - O.BETA2)
                self.p[k].sub_(alpha * self.m[k] / (self.v[k].sqrt() + O.ADAM_EPS))
        self.step = t
        return loss

    # -- cfg-E: first-order MAML (oracle/lstm_oracle.py maml_step) ------------------
    def _clip_scale(self, g, slices_sq):
        sq = 0.0
        for k in self.names:
            sq += slices_sq if (k == 'embedding' and self.clip_norm_mode == 'tf1_slices') \
                else float((g[k].double() ** 2).sum())
        clip = float(self.config['max_grad_norm'])
        return clip / max(np.sqrt(sq), clip)

    def maml_query_grads(self, support, query, inner_steps, inner_lr):
        """autograd with the inner steps DETACHED: the query gradient is taken w.r.t. the adapted parameters"""
        theta = {k: v.detach().clone() for k, v in self.p.items()}
        Xs, Ys = O.tokens_to_input_and_target(support, self.d['start'])
        for _ in range(int(inner_steps)):
            _, g, sq = self.grads_xy(Xs, Ys)
            scale = self._clip_scale(g, sq)
            with torch.no_grad():
                for k in self.names:
                    self.p[k].sub_(inner_lr * scale * g[k])
        Xq, Yq = O.tokens_to_input_and_target(query, self.d['start'])
        loss, g, sq = self.grads_xy(Xq, Yq)
        g = {k: v.clone() for k, v in g.items()}
        with torch.no_grad():
            for k in self.names:
                self.p[k].copy_(theta[k])
        return loss, g, sq

    # -- plugin-level ---------------------------------------------------------
    def train(self, support, query):
        X, Y = O.train_xy(support, query, self.d['start'])
        return self.train_xy(X, Y)

    def eval(self, query):
        X, Y = O.eval_xy(query, self.d['start'])
        return self.eval_xy(X, Y)

    def numpy_params(self):
        return {k: v.detach().numpy().copy() for k, v in self.p.items()}
