"""Plugin contract of `models/` and the token -> (input, target) helper.

Same surface as /root/reference/src/models/base_model.py:4-86 -- `train.train` resolves
`config['model_module_name']`.`config['model_class_name']`, constructs it with the merged
config dict and only ever calls the members below.
"""
import numpy as np


class BaseModel(object):
    """name / train / eval / sample / save / recover_or_init."""

    def __init__(self, config):
        self._config = config

    @property
    def name(self):
        return self._config['name']

    def train(self, episode):
        """One update on `episode` (support + query); returns the step's mean NLL."""
        raise NotImplementedError()

    def eval(self, episode):
        """Mean NLL of the episode's query set; no state change."""
        raise NotImplementedError()

    def sample(self, support_set, num):
        """`num` token ids conditioned on `support_set` ([K, T] ints)."""
        raise NotImplementedError()

    def save(self, checkpt_path):
        """Write the current parameters under `checkpt_path`."""
        raise NotImplementedError()

    def recover_or_init(self, init_path):
        """Load parameters from `init_path` if it holds any, initialise the rest."""
        raise NotImplementedError()


def flatten_first_two_dims(token_array):
    """[B, S, N] -> [B*S, N] (a view when the input is contiguous)."""
    token_array = np.asarray(token_array)
    b, s, n = token_array.shape
    return token_array.reshape(b * s, n)


def convert_tokens_to_input_and_target(token_array, start_word=None):
    """Language-model input/target pair for every song of a [B, S, N] token array.

    With a start word: target = the song, input = [start_word, song[:-1]].
    Without: input = song[:-1], target = song[1:].  (base_model.py:63-86; the HIP
    path never materialises these -- it folds the shift into its token-prep kernel --
    this host version exists for plugins that want it and for the parity tests.)
    """
    songs = flatten_first_two_dims(token_array)
    if start_word is None:
        return songs[:, :-1], songs[:, 1:].copy()
    inputs = np.empty_like(songs)
    inputs[:, 0] = start_word
    inputs[:, 1:] = songs[:, :-1]
    return inputs, songs.copy()
