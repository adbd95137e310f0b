#!/usr/bin/env python3
"""Golden vectors for the MIDI event tokenizer, produced by IMPORTING the reference (this container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_midi.py

pretty_midi is absent, so a stub module with plain container classes is injected; the reference's pipeline only
touches attributes.  Output: g5_midi.json = random duck-typed songs (several
instruments per General-MIDI family, drums, sustain-pedal events, zero-length and overlapping notes) with the
reference's `MIDILoader.tokenize` ids, and token streams with the notes the reference's `detokenize` builds.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class _Bag(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def stub_pretty_midi():
    pm = types.ModuleType('pretty_midi')

    class PrettyMIDI(object):
        def __init__(self, *a, **k):
            self.instruments = []

    class Instrument(object):
        def __init__(self, program=0, is_drum=False):
            self.program, self.is_drum, self.notes, self.control_changes = program, is_drum, [], []

    class Note(object):
        def __init__(self, velocity, pitch, start, end):
            self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    pm.PrettyMIDI, pm.Instrument, pm.Note = PrettyMIDI, Instrument, Note
    return pm


def random_song(rng, n_instr, n_notes, with_pedal):
    instruments = []
    for _ in range(n_instr):
        program = int(rng.choice([0, 1, 5, 24, 25, 26, 40, 41, 56, 73, 127]))
        is_drum = bool(rng.rand() < 0.15)
        notes, t = [], 0.0
        for _ in range(n_notes):
            t += float(rng.choice([0.0, 0.004, 0.01, 0.13, 0.5, 1.7]))
            dur = float(rng.choice([0.0, 0.003, 0.05, 0.2, 0.9, 2.5]))
            notes.append([round(t, 4), round(t + dur, 4), int(rng.choice([60, 60, 62, 64, 67, 72])), int(rng.randint(1, 128))])
        ccs = []
        if with_pedal:
            tt = 0.0
            for _ in range(int(rng.randint(1, 6))):
                tt += float(rng.uniform(0.05, 2.0))
                ccs.append([int(rng.choice([64, 64, 64, 7])), int(rng.choice([0, 30, 64, 100, 127])), round(tt, 4)])
        instruments.append(dict(program=program, is_drum=is_drum, notes=notes, control_changes=ccs))
    return dict(instruments=instruments)


def build(song):
    return _Bag(instruments=[_Bag(program=i['program'], is_drum=i['is_drum'],
                                  notes=[_Bag(start=a, end=b, pitch=p, velocity=v) for a, b, p, v in i['notes']],
                                  control_changes=[_Bag(number=n, value=v, time=t) for n, v, t in i['control_changes']])
                             for i in song['instruments']])


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, '/root/reference/src')
    sys.modules['pretty_midi'] = stub_pretty_midi()
    from data.midi_loader import MIDILoader          # noqa: E402  (reference code)
    loader = MIDILoader(64)
    rng = np.random.RandomState(7)
    cases = []
    for k in range(24):
        song = random_song(rng, n_instr=int(rng.randint(1, 5)), n_notes=int(rng.randint(1, 14)), with_pedal=k % 2 == 1)
        tokens = [int(t) for t in loader.tokenize(build(song))]
        assert all(0 <= t < loader.get_num_tokens() for t in tokens)
        cases.append(dict(song=song, tokens=tokens))
    cases.append(dict(song=dict(instruments=[]), tokens=[int(t) for t in loader.tokenize(build(dict(instruments=[])))]))
    decode = []
    streams = [c['tokens'] for c in cases[:8]] + [[int(t) for t in rng.randint(0, 4708, size=60)] for _ in range(6)]
    for toks in streams:
        midi = loader.detokenize(np.asarray(toks, np.int32))
        decode.append(dict(tokens=toks, instruments=[dict(program=int(i.program),
                      notes=[[float(n.start), float(n.end), int(n.pitch), int(n.velocity)] for n in i.notes])
                      for i in midi.instruments]))
    with open(os.path.join(HERE, 'g5_midi.json'), 'w') as f:
        json.dump(dict(vocab=loader.get_num_tokens(), tokenize=cases, detokenize=decode), f)
    print('cases', len(cases), 'decode', len(decode), 'tokens', sum(len(c['tokens']) for c in cases))


if __name__ == '__main__':
    main()
