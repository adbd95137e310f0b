"""MAML-style few-shot LSTM language model (BASELINE.json configs[4], "cfg-E") -- MI355X-native plugin.

The reference ships no meta-learning model; its READING_LIST.md:5-7 names MAML ("Model-agnostic meta-learning for fast
adaptation of deep networks") as the direction the project was heading.  This plugin keeps the models/ plugin API
(src/models/base_model.py:4-54) and the LSTM-baseline architecture and config keys, and changes what an episode means
(semantics in DESIGN.md "cfg-E"; CPU oracle oracle/lstm_oracle.py maml_step / maml_eval), first order:

  train(episode)  theta' = theta after `inner_steps` steps of clipped SGD (`inner_lr`) on the SUPPORT set; the gradient
                  of the QUERY set's mean NLL at theta' is applied to theta with the baseline's clip + Adam; returns the
                  query NLL at theta' (before the outer update).  Episode-parallel: the inner loop is per rank and needs no
                  communication, the query gradients are all-reduced exactly like a baseline step's.
  eval(episode)   adapt on the support set, mean NLL of the query set at theta', then theta is restored.
  sample(s, num)  greedy decode at theta (the support set is not used, as in the baseline).

Extra config keys: inner_steps (default 1), inner_lr (default 0.1).
"""
from models.lstm_baseline import LSTMBaseline


class MAMLLSTM(LSTMBaseline):
    def __init__(self, config):
        super(MAMLLSTM, self).__init__(config)
        self._inner_steps = int(config.get('inner_steps', 1))
        self._inner_lr = float(config.get('inner_lr', 0.1))
        if self._inner_steps < 0 or self._inner_lr < 0:
            raise RuntimeError('inner_steps and inner_lr must be >= 0')

    def train(self, episode):
        self._require_init()
        loss = self._parallel.train_step(self._tokens(episode.support, 3), self._tokens(episode.query, 3),
                                         maml=(self._inner_steps, self._inner_lr))
        self._log_scalar('Train/loss', loss, self._train_calls)
        self._train_calls += 1
        return loss

    # train.train's fast path (episodes as row indices into the split's device-resident token table, losses left on the device): the
    # inherited LSTMBaseline.train_indexed would run the PLAIN baseline step -- inner loop skipped, training and evaluation
    # silently disagreeing.  The table lives on the device like the baseline's (inherited attach_table): the episode's
    # rows are gathered there once and both passes of the step read them in place (fsmg_maml_step_indexed).
    def train_indexed(self, split, support_idx, query_idx, want_loss=False):
        import numpy as np
        self._require_init()
        loss = self._parallel.train_step(np.ascontiguousarray(support_idx, dtype=np.int32),
                                         np.ascontiguousarray(query_idx, dtype=np.int32), want_loss=want_loss,
                                         table=self.TABLE_IDS[split], maml=(self._inner_steps, self._inner_lr))
        if want_loss:
            self._log_scalar('Train/loss', loss, self._train_calls)
        self._train_calls += 1
        return loss

    def eval(self, episode):
        self._require_init()
        nll = self._model.maml_eval(self._tokens(episode.support, 3), self._tokens(episode.query, 3),
                                    self._inner_steps, self._inner_lr)
        self._log_scalar('Eval/Avg_NLL', nll, self._eval_calls)
        self._eval_calls += 1
        return nll

    def eval_many(self, episodes):
        """adaptation is per episode, so there is nothing to batch: one maml_eval per episode"""
        return [self.eval(e) for e in episodes]
