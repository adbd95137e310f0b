"""Unigram baseline: add-one-smoothed token counts over the meta-training stream.

Behaviour of /root/reference/src/models/unigram_model.py:8-78 (a TensorFlow scatter_add histogram), kept selectable through
`unigram.yaml`.  The counts live on the MI355X behind the C-ABI (`fsmg_unigram_*`,
include/fsmg.h: histogram with integer atomics, gather / reduce_sum / -mean(log) in one small kernel); like the LSTM
plugin there is no CPU fallback -- without a gfx950 device the constructor raises.

  train(episode)  words = tokens[:, :-1] of the support rows then the query rows (convert_tokens_to_input_and_target
                  WITHOUT a start word, reference :41-49): counts[w] += 1 per occurrence; returns the mean negative log
                  probability of those same words (reference :51-56).  The reference fetches the scatter_add and the loss in
                  one sess.run with no control dependency, so whether its loss sees the counts before or after the update is
                  not defined; here it is computed BEFORE, like LSTMBaseline.train's pre-update loss.
  eval(episode)   words = tokens[:, 1:] of the QUERY rows only (reference :58-69); support set ignored; no state change.
  sample(s, num)  the most frequent word, `num` times (reference :71-78).

`host_unigram_nll / host_unigram_update` restate the same three lines of arithmetic in numpy: the GPU test's checker and the
known-answer CPU test -- not a product path.
"""
import os

import numpy as np

from models.base_model import BaseModel, convert_tokens_to_input_and_target

ALPHA = 1.0            # add-one smoothing (reference :24)


def host_unigram_nll(counts, words):
    """-mean(log(gather(counts, words) / reduce_sum(counts))) in float32 (reference :35-37)"""
    prob = counts[words] / counts.sum(dtype=np.float32)
    return float(-np.mean(np.log(prob), dtype=np.float32))


def host_unigram_update(counts, words):
    """scatter_add of ones (reference :33), in place"""
    np.add.at(counts, words, np.float32(1.0))


def train_words(episode):
    x_sup, _ = convert_tokens_to_input_and_target(episode.support)
    x_qry, _ = convert_tokens_to_input_and_target(episode.query)
    return np.concatenate([x_sup, x_qry]).ravel().astype(np.int32)


def eval_words(episode):
    _, y = convert_tokens_to_input_and_target(episode.query)
    return y.ravel().astype(np.int32)


class UnigramModel(BaseModel):
    def __init__(self, config):
        super(UnigramModel, self).__init__(config)
        from fsmg.binding import FsmgUnigram
        self.engine = FsmgUnigram(int(config['input_size']), device=int(config.get('device', 0)))    # word_count = alpha (reference :27-30)

    def train(self, episode):
        return self.engine.train(train_words(episode))

    def eval(self, episode):
        return self.engine.nll(eval_words(episode))

    def sample(self, support_set, num):
        return [self.engine.argmax()] * int(num)

    def save(self, checkpt_path):
        directory = os.path.join(checkpt_path, self.name)
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, self.name + '.npy'), self.engine.get_counts())

    def recover_or_init(self, init_path):
        path = os.path.join(init_path or '', self.name, self.name + '.npy')
        if init_path and os.path.isfile(path):
            self.engine.set_counts(np.load(path).astype(np.float32))
