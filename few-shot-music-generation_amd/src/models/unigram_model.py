"""Unigram baseline: add-one-smoothed token counts.

Keeps `unigram.yaml` selectable (/root/reference/src/models/unigram_model.py:8-78 is a
TensorFlow scatter_add histogram; SURVEY.md section 8 row f-4 ranks it outside the hot path).
It is a few-kB histogram, so it stays on the host in numpy -- it is NOT part of the HIP path and
no parity or performance claim is made for it.
"""
import os

import numpy as np

from models.base_model import BaseModel, flatten_first_two_dims


class UnigramModel(BaseModel):
    def __init__(self, config):
        super(UnigramModel, self).__init__(config)
        self._counts = np.ones(int(config['input_size']), np.float64)       # add-one smoothing

    def train(self, episode):
        loss = self.eval(episode)
        np.add.at(self._counts, flatten_first_two_dims(episode.support).ravel(), 1.0)
        np.add.at(self._counts, flatten_first_two_dims(episode.query).ravel(), 1.0)
        return loss

    def eval(self, episode):
        tokens = flatten_first_two_dims(episode.query).ravel()
        return float(-np.mean(np.log(self._counts[tokens] / self._counts.sum())))

    def sample(self, support_set, num):
        return [int(np.argmax(self._counts))] * int(num)

    def save(self, checkpt_path):
        directory = os.path.join(checkpt_path, self.name)
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, self.name + '.npy'), self._counts)

    def recover_or_init(self, init_path):
        path = os.path.join(init_path or '', self.name, self.name + '.npy')
        if init_path and os.path.isfile(path):
            self._counts = np.load(path)
