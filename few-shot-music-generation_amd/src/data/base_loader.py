"""Per-song token loading: the on-disk token format of the reference.

Mirrors the behaviour of the reference `Loader` (/root/reference/src/data/base_loader.py:12-64):
a song file `<path>` has a pre-tokenised sidecar `<path>.<max_len>.npy` (int32
[max_len]); when the sidecar is missing the song is read, tokenised, truncated /
ZERO-padded to max_len (padding is not masked downstream and id 0 is also a real
token -- SURVEY.md Q6) and the sidecar is written.
"""
import os

import numpy as np

_VALIDATION_ERRORS = (OSError, KeyError, EOFError, IndexError, ValueError, IOError)


class Loader(object):
    """Turns a song file into a fixed-length row of token ids."""

    def __init__(self, max_len, dtype=np.int32, persist=True):
        self.max_len = int(max_len)
        self.dtype = dtype
        self.persist = persist

    # -- to be provided by the concrete loader --------------------------------
    def is_song(self, filepath):
        raise NotImplementedError

    def read(self, filepath):
        raise NotImplementedError

    def tokenize(self, data):
        raise NotImplementedError

    def detokenize(self, numpy_data):
        raise NotImplementedError

    def get_num_tokens(self):
        raise NotImplementedError

    # -- shared ------------------------------------------------------------------
    def sidecar_path(self, filepath):
        return '%s.%s.npy' % (filepath, self.max_len)

    def validate(self, filepath):
        """A song is valid iff it loads (base_loader.py:35-50)."""
        try:
            self.load(filepath)
        except _VALIDATION_ERRORS:
            return False
        return True

    def load(self, filepath):
        """-> int32 [max_len] (base_loader.py:52-64)."""
        sidecar = self.sidecar_path(filepath)
        if self.persist and os.path.isfile(sidecar):
            return np.load(sidecar).astype(self.dtype)
        ids = self.tokenize(self.read(filepath))
        row = np.zeros(self.max_len, dtype=self.dtype)
        n = min(self.max_len, len(ids))
        row[:n] = ids[:n]
        if self.persist:                        # atomically: another rank must never np.load a half-written sidecar
            tmp = '%s.%d.tmp.npy' % (sidecar, os.getpid())
            np.save(tmp, row)
            os.replace(tmp, sidecar)
        return row
