"""MIDI event-token loader -- vocabulary and pre-tokenised rows only.

The reference's raw-MIDI tokenizer (/root/reference/src/data/midi_loader.py:62-399)
needs `pretty_midi`, which is absent here, and is offline preprocessing outside
the hot path.  What the hot path needs is kept:
the vocabulary size (midi_loader.py:53-60: 16 instrument families x 128 pitches x
{on, off} + 32 velocity bins x 16 families + 100 time-shift steps = 4708) and the
`<song>.mid.<max_len>.npy` sidecars written by the reference's loader.
"""
import numpy as np

from data.base_loader import Loader

NUM_FAMILIES = 16
NUM_PITCHES = 128
NUM_VELOCITY_BINS = 32
NUM_TIME_SHIFTS = 100


class MIDILoader(Loader):
    def __init__(self, max_len, dtype=np.int32, persist=True):
        super(MIDILoader, self).__init__(max_len, dtype=dtype, persist=persist)

    def is_song(self, filepath):
        return filepath.endswith('.mid')

    def get_num_tokens(self):
        return NUM_FAMILIES * NUM_PITCHES * 2 + NUM_VELOCITY_BINS * NUM_FAMILIES + NUM_TIME_SHIFTS

    def read(self, filepath):
        try:
            import pretty_midi
        except ImportError:
            raise OSError('raw MIDI needs pretty_midi; only pre-tokenised %s sidecars are supported here'
                          % self.sidecar_path(filepath))
        return pretty_midi.PrettyMIDI(filepath)

    def tokenize(self, midi):
        raise OSError('raw-MIDI event tokenisation is out of scope (SURVEY.md 8 f-3)')

    def detokenize(self, numpy_data):
        """No MIDI writer here: return the event ids as text so sample dumps still work."""
        return ' '.join(str(int(t)) for t in numpy_data)
