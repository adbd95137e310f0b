"""LSTM language-model baseline -- MI355X-native plugin.

Drop-in for /root/reference/src/models/lstm_baseline.py: same module path and class name
(`model_module_name: models.lstm_baseline`, `model_class_name: LSTMBaseline`), same config
keys, same six plugin members with the same meaning -- but the TensorFlow graph and session
are replaced by libfsmg (hand-written gfx950 kernels behind include/fsmg.h):

  train(episode)  support + query flattened support-first, one clip+Adam update, returns the
                  mean NLL under the PRE-update weights      (reference :89-113)
  eval(episode)   query-only mean NLL, no state change       (reference :115-133)
  sample(s, num)  greedy argmax decode from the start word; the support set is ignored, as in
                  the reference                              (reference :135-156)

Optional config keys beyond the reference's: device, clip_norm_mode ('tf1_slices' | 'dense'),
max_sequences, use_graph, gemm / schedule / recurrence / dp_split_backward (fsmg_config), dp_exchange ('torch': the
all-reduce is issued through torch.distributed; 'library': libfsmg issues the RCCL calls itself).  When torch.distributed is initialised, train() runs episode-parallel
(one episode per rank, one gradient all-reduce per step, see fsmg/dist.py).
"""
import numpy as np

from fsmg.dist import EpisodeParallel
from models.hip_model import HIPModel


class LSTMBaseline(HIPModel):
    def __init__(self, config):
        for key in ('name', 'input_size', 'max_len', 'embedding_size', 'hidden_size', 'n_layers',
                    'lr', 'max_grad_norm', 'n_decay'):
            if key not in config:
                raise RuntimeError('required config key "%s" not found' % key)
        super(LSTMBaseline, self).__init__(config)
        self._start_word = int(config['input_size'])
        self._time_steps = int(config['max_len'])
        self._parallel = EpisodeParallel(self)
        if config.get('dp_exchange', 'torch') == 'library' and self._parallel.world > 1:
            self.attach_library_comm()

    def recover_or_init(self, init_path):
        super(LSTMBaseline, self).recover_or_init(init_path)
        if self._parallel.world > 1:                     # replicas start from rank 0's state
            self._parallel.broadcast_parameters(self._arena)

    @staticmethod
    def _tokens(arr, ndim):
        a = np.ascontiguousarray(arr, dtype=np.int32)
        if a.ndim != ndim:
            raise ValueError('expected a %d-d token array, got shape %r' % (ndim, a.shape))
        return a

    def train(self, episode):
        self._require_init()
        loss = self._parallel.train_step(self._tokens(episode.support, 3), self._tokens(episode.query, 3))
        self._log_scalar('Train/loss', loss, self._train_calls)
        self._train_calls += 1
        return loss

    # -- fast path of train.train: episodes as row indices into a device-resident split table, losses left on the device --
    TABLE_IDS = {'train': 0, 'val': 1, 'test': 2}

    def attach_table(self, split, table):
        """Upload the packed [n_songs, max_len] int32 token table of a split (data.dataset.Dataset.token_table) once; episodes
        of that split can then be given as row indices (train_indexed)."""
        self._model.upload_table(self.TABLE_IDS[split], table)

    def train_indexed(self, split, support_idx, query_idx, want_loss=False):
        """Same step as train(episode) for the episode table[support_idx], table[query_idx]; with want_loss=False nothing is
        read back (the loss goes to the device ring: recent_losses) and the host does not wait for the GPU."""
        self._require_init()
        loss = self._parallel.train_step(np.ascontiguousarray(support_idx, dtype=np.int32),
                                         np.ascontiguousarray(query_idx, dtype=np.int32), want_loss=want_loss,
                                         table=self.TABLE_IDS[split])
        if want_loss:
            self._log_scalar('Train/loss', loss, self._train_calls)
        self._train_calls += 1
        return loss

    def recent_losses(self, n):
        """the last n (<= 1024) train losses, oldest first; synchronises"""
        return self._model.read_losses(int(n))

    def global_step(self):
        """train steps the device has really applied (a skipped step does not count); synchronises"""
        return self._model.step

    def log_deferred_losses(self, losses):
        """Train/loss scalars of steps that ran with want_loss=False, written when train.train folds them in"""
        # labelled by the step the device really applied (global_step counts those; a skipped step leaves no loss behind), so the
        # window's last loss carries the current global_step - 1
        first = self._model.step - len(losses)
        for i, loss in enumerate(losses):
            self._log_scalar('Train/loss', float(loss), max(first + i, 0))

    def eval(self, episode):
        self._require_init()
        nll = self._model.eval_step(self._tokens(episode.query, 3))
        self._log_scalar('Eval/Avg_NLL', nll, self._eval_calls)
        self._eval_calls += 1
        return nll

    def eval_many(self, episodes):
        """[model.eval(e) for e in episodes] in one device pass (episodes must share N, Q)."""
        self._require_init()
        queries = np.stack([self._tokens(e.query, 3) for e in episodes])
        nlls = self._model.eval_batch(queries)
        for nll in nlls:
            self._log_scalar('Eval/Avg_NLL', float(nll), self._eval_calls)
            self._eval_calls += 1
        return [float(x) for x in nlls]

    def sample(self, support_set, num):
        self._require_init()
        return self._model.sample(int(num))
