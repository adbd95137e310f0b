"""MIDI event-token loader.

Behaviour of the reference `MIDILoader` (/root/reference/src/data/midi_loader.py:37-399), re-implemented on
plain records so that it needs `pretty_midi` only to PARSE a raw `.mid` file (`read`); everything after that
-- sustain-pedal handling, 10 ms quantisation, drum removal, same-family pitch-clash resolution, the event
list and the token ids -- works on any object with the same attributes (duck typing):

    song.instruments[i].program / .is_drum / .notes[j].(start, end, pitch, velocity)
                                             / .control_changes[j].(number, value, time)

Token layout (midi_loader.py:53-88), vocabulary 16*128*2 + 32*16 + 100 = 4708:
    NOTE_ON   family*128 + pitch               family = program // 8 + 1  (1-based, see below)
    NOTE_OFF  2048 + family*128 + pitch
    VELOCITY  4096 + 32*family + bin           bin = (velocity - 1) // 4 + 1   (1..32)
    TIME      4608 + steps - 1                 steps in 1..100 (10 ms units)
Quirks kept on purpose: `tokenize` uses the 1-based family while `detokenize` decodes a
0-based one, so the id ranges overlap; ids stay < 4708 and the model is unaffected.  The reference's attempt
to drop zero-length notes under sustain (`midi_notes.remove(note)`, :336-339) compares a tuple with a note and
never removes anything; neither do we.  Unlike the reference, the caller's note objects are not mutated.

`detokenize` returns a `MidiSong` whose `.write(path)` emits a Standard MIDI File with a built-in writer
(the reference builds a `pretty_midi.PrettyMIDI`, :114-128): same instruments, notes, times and velocities.
"""
import struct

import numpy as np

from data.base_loader import Loader

NUM_FAMILIES = 16
NUM_PITCHES = 128
NUM_VELOCITY_BINS = 32
MAX_SHIFT_STEPS = 100
PROGRAMS_PER_FAMILY = 8
VELOCITY_BIN_SIZE = 4            # ceil(127 / 32)
STEPS_PER_SECOND = 100
SUSTAIN_CONTROL = 64

OFF_NOTE_OFF = NUM_FAMILIES * NUM_PITCHES
OFF_VELOCITY = 2 * NUM_FAMILIES * NUM_PITCHES
OFF_TIME = OFF_VELOCITY + NUM_VELOCITY_BINS * NUM_FAMILIES


class _Note(object):
    """Working copy of one note (times in seconds until quantised)."""
    __slots__ = ('start', 'end', 'pitch', 'velocity', 'program', 'instrument', 'is_drum')

    def __init__(self, src, program, instrument, is_drum):
        self.start, self.end = src.start, src.end
        self.pitch, self.velocity = int(src.pitch), int(src.velocity)
        self.program, self.instrument, self.is_drum = int(program), instrument, bool(is_drum)


def collect_notes(song):
    """All notes of all instruments, in instrument then note order (midi_loader.py:384-399)."""
    return [_Note(n, ins.program, i, ins.is_drum) for i, ins in enumerate(song.instruments) for n in ins.notes]


def apply_sustain(notes, song):
    """Sustain pedal (controller 64, >= 64 on): a sustained note rings until the pedal is released, until the
    same pitch is struck again on that instrument, or until the last event of the song
    (midi_loader.py:284-363).  Events are ordered by time only (stable: note-ons, note-offs, pedal events)."""
    ON, OFF, NOTE_ON, NOTE_OFF = 0, 1, 2, 3
    events = [(n.start, NOTE_ON, n.instrument, n) for n in notes]
    events += [(n.end, NOTE_OFF, n.instrument, n) for n in notes]
    for i, ins in enumerate(song.instruments):
        for cc in getattr(ins, 'control_changes', ()):
            if cc.number == SUSTAIN_CONTROL:
                events.append((cc.time, ON if cc.value >= 64 else OFF, i, None))
    events.sort(key=lambda e: e[0])

    held = {}            # instrument -> notes currently sounding
    pedal = {}           # instrument -> pedal down?
    now = 0
    for now, kind, ins, note in events:
        ringing = held.setdefault(ins, [])
        if kind == ON:
            pedal[ins] = True
        elif kind == OFF:
            pedal[ins] = False
            keep = []
            for n in ringing:
                if n.end < now:
                    n.end = now          # key already released: the pedal was holding it
                else:
                    keep.append(n)
            held[ins] = keep
        elif kind == NOTE_ON:
            if pedal.get(ins, False):
                keep = []
                for n in ringing:
                    if n.pitch == note.pitch:
                        n.end = now      # re-struck under the pedal: the old one stops here
                    else:
                        keep.append(n)
                ringing = held[ins] = keep
            ringing.append(note)
        elif not pedal.get(ins, False) and note in ringing:
            ringing.remove(note)
    for ringing in held.values():
        for n in ringing:
            n.end = now                  # still held when the song's event list ends
    return notes


def quantize(notes):
    """seconds -> 10 ms steps, round half up; zero-length notes last one step (midi_loader.py:258-281)."""
    for n in notes:
        n.start = int(n.start * STEPS_PER_SECOND + 0.5)
        n.end = int(n.end * STEPS_PER_SECOND + 0.5)
        if n.end == n.start:
            n.end += 1
    return notes


def family_of(program):
    return program // PROGRAMS_PER_FAMILY + 1


def resolve_pitch_clashes(notes):
    """Instruments of one General-MIDI family are merged; when two of them sound the same pitch at once the first
    note finishes and only the remainder of the second is kept (midi_loader.py:131-184)."""
    ordered = sorted(notes, key=lambda n: (n.start, n.end, n.program))
    sounding = {}        # family -> [(pitch, end)]
    kept = []
    for n in ordered:
        fam = family_of(n.program)
        active = [(p, e) for p, e in sounding.get(fam, ()) if e > n.start]
        latest = max([e for p, e in active if p == n.pitch], default=0)
        if latest >= n.end:
            sounding[fam] = active
            continue                     # completely covered by a sounding note
        n.start = max(n.start, latest)
        active.append((n.pitch, n.end))
        sounding[fam] = active
        kept.append(n)
    return kept


def event_list(notes):
    """Notes -> (kind, value, family) events: TIME shifts of at most 100 steps, a VELOCITY event whenever a
    family's velocity bin changes, NOTE_ON / NOTE_OFF (midi_loader.py:200-255)."""
    marks = sorted([(n.start, i, n.program, False) for i, n in enumerate(notes)] +
                   [(n.end, i, n.program, True) for i, n in enumerate(notes)])
    events, now, bins = [], 0, {}
    for step, i, program, is_off in marks:
        if step > now:
            while step > now + MAX_SHIFT_STEPS:
                events.append(('time', MAX_SHIFT_STEPS, 0))
                now += MAX_SHIFT_STEPS
            events.append(('time', step - now, 0))
            now = step
        n = notes[i]
        fam = family_of(program)
        if is_off:
            events.append(('off', n.pitch, fam))
            continue
        vbin = (n.velocity - 1) // VELOCITY_BIN_SIZE + 1
        if bins.get(fam, 0) != vbin:
            bins[fam] = vbin
            events.append(('velocity', vbin, fam))
        events.append(('on', n.pitch, fam))
    return events


def events_to_tokens(events):
    ids = []
    for kind, value, fam in events:
        if kind == 'on':
            ids.append(fam * NUM_PITCHES + value)
        elif kind == 'off':
            ids.append(OFF_NOTE_OFF + fam * NUM_PITCHES + value)
        elif kind == 'velocity':
            ids.append(OFF_VELOCITY + NUM_VELOCITY_BINS * fam + value)
        else:
            ids.append(OFF_TIME + value - 1)
    return ids


def tokenize_song(song):
    """The whole pipeline of MIDILoader.tokenize (midi_loader.py:62-88)."""
    notes = quantize(apply_sustain(collect_notes(song), song))
    notes = [n for n in notes if not n.is_drum]
    return events_to_tokens(event_list(resolve_pitch_clashes(notes)))


# ------------------------------------------------------------------------------------------- decoding
class MidiSong(object):
    """What detokenize produces: per instrument class a program number and (start_s, end_s, pitch, velocity) notes."""

    def __init__(self):
        self.instruments = []            # [(program, [(start, end, pitch, velocity), ...])]

    def write(self, path, ticks_per_beat=480, tempo_us=500000):
        """Minimal Standard MIDI File (format 1): a tempo track + one track per instrument."""
        def varlen(v):
            out = [v & 0x7F]
            v >>= 7
            while v:
                out.append((v & 0x7F) | 0x80)
                v >>= 7
            return bytes(reversed(out))

        def chunk(tag, body):
            return tag + struct.pack('>I', len(body)) + body

        ticks_per_s = ticks_per_beat * 1e6 / tempo_us
        tracks = [b'\x00\xff\x51\x03' + struct.pack('>I', tempo_us)[1:] + b'\x00\xff\x2f\x00']
        for index, (program, notes) in enumerate(self.instruments):
            channel = index if index < 9 else index + 1            # channel 10 (index 9) is percussion
            channel = min(channel, 15)
            msgs = []
            for start, end, pitch, velocity in notes:
                msgs.append((int(round(start * ticks_per_s)), 1, bytes([0x90 | channel, pitch & 0x7F, max(1, min(127, velocity))])))
                msgs.append((int(round(end * ticks_per_s)), 0, bytes([0x80 | channel, pitch & 0x7F, 0])))
            msgs.sort(key=lambda m: (m[0], m[1]))
            body, last = b'\x00' + bytes([0xC0 | channel, program & 0x7F]), 0
            for tick, _, data in msgs:
                body += varlen(tick - last) + data
                last = tick
            tracks.append(body + b'\x00\xff\x2f\x00')
        with open(path, 'wb') as f:
            f.write(chunk(b'MThd', struct.pack('>HHH', 1, len(tracks), ticks_per_beat)))
            for body in tracks:
                f.write(chunk(b'MTrk', body))


def detokenize_tokens(tokens):
    """ids -> MidiSong, decoding exactly like the reference (midi_loader.py:90-128): 0-based instrument class,
    velocity value * 4, time in 10 ms steps; a NOTE_OFF without a matching NOTE_ON is ignored."""
    now = 0
    velocity = [16] * NUM_FAMILIES
    notes = [[] for _ in range(NUM_FAMILIES)]
    started = [[None] * NUM_PITCHES for _ in range(NUM_FAMILIES)]
    for token in tokens:
        token = int(token)
        if token < OFF_NOTE_OFF:
            started[token // NUM_PITCHES][token % NUM_PITCHES] = (velocity[token // NUM_PITCHES], now)
        elif token < OFF_VELOCITY:
            cls, pitch = divmod(token - OFF_NOTE_OFF, NUM_PITCHES)
            if started[cls][pitch] is not None:
                vel, begin = started[cls][pitch]
                notes[cls].append((begin, now, pitch, vel))
                started[cls][pitch] = None
        elif token < OFF_TIME:
            cls, value = divmod(token - OFF_VELOCITY, NUM_VELOCITY_BINS)
            velocity[cls] = value
        else:
            now += token - OFF_TIME + 1
    song = MidiSong()
    for cls, cls_notes in enumerate(notes):
        if cls_notes:
            cls_notes.sort()
            song.instruments.append((cls * PROGRAMS_PER_FAMILY,
                                     [(0.01 * b, 0.01 * e, p, v * 4) for b, e, p, v in cls_notes]))
    return song


class MIDILoader(Loader):
    def __init__(self, max_len, dtype=np.int32, persist=True):
        super(MIDILoader, self).__init__(max_len, dtype=dtype, persist=persist)

    def is_song(self, filepath):
        return filepath.endswith('.mid')

    def get_num_tokens(self):
        return OFF_TIME + MAX_SHIFT_STEPS

    def read(self, filepath):
        """Parsing a raw .mid file is the one step that needs pretty_midi (absent here); datasets pre-tokenised by
        the reference (`<song>.mid.<max_len>.npy`) never reach this."""
        try:
            import pretty_midi
        except ImportError:
            # NOT an OSError: Loader.validate swallows those as "invalid song", every raw .mid would be rejected silently and
            # an empty train/val/test split persisted; the reference fails at import time, so this must abort as well
            raise RuntimeError('parsing raw MIDI needs pretty_midi; only pre-tokenised %s sidecars can be loaded without it'
                               % self.sidecar_path(filepath))
        return pretty_midi.PrettyMIDI(filepath)

    def tokenize(self, midi):
        return tokenize_song(midi)

    def detokenize(self, numpy_data):
        return detokenize_tokens(numpy_data)
