"""MIDI event tokenizer vs golden vectors produced by the reference (tests/golden/make_golden_midi.py)."""
import json
import os
import struct
from types import SimpleNamespace as NS

import numpy as np
import pytest

from data.midi_loader import MIDILoader, detokenize_tokens, tokenize_song


@pytest.fixture(scope='module')
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'g5_midi.json')))


def _song(spec):
    return NS(instruments=[NS(program=i['program'], is_drum=i['is_drum'],
                              notes=[NS(start=a, end=b, pitch=p, velocity=v) for a, b, p, v in i['notes']],
                              control_changes=[NS(number=n, value=v, time=t) for n, v, t in i['control_changes']])
                           for i in spec['instruments']])


def test_tokenize_matches_reference(gold):
    loader = MIDILoader(64)
    assert loader.get_num_tokens() == gold['vocab'] == 4708
    n_tokens = 0
    for case in gold['tokenize']:
        song = _song(case['song'])
        before = [(n.start, n.end) for i in song.instruments for n in i.notes]
        assert loader.tokenize(song) == case['tokens']
        assert [(n.start, n.end) for i in song.instruments for n in i.notes] == before   # caller's notes untouched
        n_tokens += len(case['tokens'])
    assert n_tokens > 500


def test_detokenize_matches_reference(gold):
    for case in gold['detokenize']:
        song = detokenize_tokens(np.asarray(case['tokens'], np.int32))
        got = [dict(program=p, notes=[[s, e, pi, v] for s, e, pi, v in notes]) for p, notes in song.instruments]
        assert len(got) == len(case['instruments'])
        for a, b in zip(got, case['instruments']):
            assert a['program'] == b['program']
            np.testing.assert_allclose(np.array(a['notes'], float).reshape(-1, 4), np.array(b['notes'], float).reshape(-1, 4), rtol=0, atol=1e-12)


def test_midi_file_writer_and_loader_padding(tmp_path, gold):
    toks = gold['tokenize'][0]['tokens']
    song = MIDILoader(64).detokenize(np.asarray(toks, np.int32))
    path = str(tmp_path / 'out.mid')
    song.write(path)
    blob = open(path, 'rb').read()
    assert blob[:4] == b'MThd' and struct.unpack('>HHH', blob[8:14])[1] == len(song.instruments) + 1
    assert blob.count(b'MTrk') == len(song.instruments) + 1
    # Loader.load on a pre-tokenised sidecar: truncation / zero padding to max_len like the lyrics path
    mid = tmp_path / 'a.mid'
    mid.write_bytes(b'')
    np.save(str(mid) + '.16.npy', np.arange(16, dtype=np.int32))
    assert MIDILoader(16).load(str(mid)).tolist() == list(range(16))
    with pytest.raises(RuntimeError):
        MIDILoader(8).load(str(mid))            # no sidecar for max_len=8 and no pretty_midi to parse with
    with pytest.raises(RuntimeError):           # ... and validate() must NOT swallow the missing dependency as "invalid song"
        MIDILoader(8).validate(str(mid))        # (that would persist an empty split; the reference fails at import time)
    assert MIDILoader(8).is_song('x.mid')
