"""Unigram baseline: add-one-smoothed token counts over the meta-training stream.

Behaviour of /root/reference/src/models/unigram_model.py:8-78 (a TensorFlow scatter_add histogram), kept selectable through
`unigram.yaml`.  It is a few-kB histogram, so it stays on the host in numpy float32 (the
reference's variable dtype) -- it is not part of the HIP path and no performance claim is made for it.

  train(episode)  words = tokens[:, :-1] of the support rows then the query rows (convert_tokens_to_input_and_target
                  WITHOUT a start word, reference :41-49): counts[w] += 1 per occurrence; returns the mean negative log
                  probability of those same words (reference :51-56).  The reference fetches the scatter_add and the loss in
                  one sess.run with no control dependency, so whether its loss sees the counts before or after the update is
                  not defined; here it is computed BEFORE, like LSTMBaseline.train's pre-update loss.
  eval(episode)   words = tokens[:, 1:] of the QUERY rows only (reference :58-69); support set ignored; no state change.
  sample(s, num)  the most frequent word, `num` times (reference :71-78).
"""
import os

import numpy as np

from models.base_model import BaseModel, convert_tokens_to_input_and_target

ALPHA = 1.0            # add-one smoothing (reference :24)


class UnigramModel(BaseModel):
    def __init__(self, config):
        super(UnigramModel, self).__init__(config)
        self._counts = np.full(int(config['input_size']), ALPHA, np.float32)       # word_count variable (reference :27-30)

    def _avg_neg_log(self, words):
        prob = self._counts[words] / self._counts.sum(dtype=np.float32)            # gather / reduce_sum (reference :35-36)
        return float(-np.mean(np.log(prob), dtype=np.float32))                     # (reference :37)

    def train(self, episode):
        x_sup, _ = convert_tokens_to_input_and_target(episode.support)
        x_qry, _ = convert_tokens_to_input_and_target(episode.query)
        words = np.concatenate([x_sup, x_qry]).ravel()
        loss = self._avg_neg_log(words)
        np.add.at(self._counts, words, np.float32(1.0))                            # scatter_add (reference :33)
        return loss

    def eval(self, episode):
        _, y = convert_tokens_to_input_and_target(episode.query)
        return self._avg_neg_log(y.ravel())

    def sample(self, support_set, num):
        return [int(np.argmax(self._counts))] * int(num)

    def save(self, checkpt_path):
        directory = os.path.join(checkpt_path, self.name)
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, self.name + '.npy'), self._counts)

    def recover_or_init(self, init_path):
        path = os.path.join(init_path or '', self.name, self.name + '.npy')
        if init_path and os.path.isfile(path):
            self._counts = np.load(path).astype(np.float32)
