"""Artist/song index, train/val/test artist splits, persisted metadata.

On-disk formats follow the reference (/root/reference/src/data/dataset.py:85-199) so
datasets pre-processed by it are read as they are:
  <root>/<artist>/<song>                         song files (.txt / .mid)
  <root>/<artist>/<song>.<max_len>.npy           int32 token rows (base_loader)
  <root>/few_shot_metadata_<dataset>_<max_len>/  valid_songs.csv  (`quote(artist),quote(song)` lines)
                                                 {train,val,test}.csv (one artist per line, no trailing newline)
                                                 word_ids.csv (lyrics vocabulary)
Deliberate deviations:
  Q3  artist scan order is sorted(os.listdir) and per-artist song order is the
      order of first appearance (valid_songs.csv order, else sorted) instead of
      hash-dependent list(set(..)); `song_order=` pins an explicit order.
  Q4  metadata paths are joined once (the reference joins the directory twice,
      which only works for an absolute dataset_path -- train.py:52 makes it so).
The whole split is also available as one packed int32 [n_songs, max_len] table
(`token_table`) so an episode is an index gather.
"""
import logging
import os
import time
from urllib.parse import quote, unquote

import numpy as np

log = logging.getLogger('few-shot')
logging.basicConfig(level=logging.INFO)


class Metadata(object):
    """Small append-only text files kept next to the dataset."""

    def __init__(self, root, name):
        self.dir = os.path.join(root, name)
        self.open_files = {}
        os.makedirs(self.dir, exist_ok=True)

    def path(self, filename):
        return os.path.join(self.dir, filename)

    def exists(self, filename):
        return os.path.exists(self.path(filename))

    def lines(self, filename):
        if self.exists(filename):
            with open(self.path(filename), 'r') as f:
                for line in f:
                    yield line

    def write(self, filename, line):
        if filename not in self.open_files:
            self.open_files[filename] = open(self.path(filename), 'a')
        self.open_files[filename].write(line)

    def write_whole(self, filename, text):
        """a complete file in one atomic step (tmp + rename): a concurrent reader sees the old file or the new one, never half"""
        tmp = '%s.%d.tmp' % (self.path(filename), os.getpid())
        with open(tmp, 'w') as f:
            f.write(text)
        os.replace(tmp, self.path(filename))

    def close(self):
        for f in self.open_files.values():
            f.close()
        self.open_files = {}


class ArtistDataset(object):
    def __init__(self, artist, songs):
        self.name = artist
        self.songs = list(songs)

    def __len__(self):
        return len(self.songs)

    def __getitem__(self, index):
        return self.songs[index]


class ProgressLogger(object):
    """At most one progress line per second while validating songs."""

    def __init__(self, num_dirs):
        self.num_dirs = max(1, num_dirs)
        self.last_log = 0.0
        self.last_percent = None

    def maybe_log(self, index):
        now = time.time()
        if now - self.last_log < 1:
            return
        percent = '%.2f' % (100.0 * index / self.num_dirs)
        if percent != self.last_percent:
            self.last_percent = percent
            self.last_log = now
            log.info('Preprocessing data. %s%%' % percent)


class Dataset(object):
    """One split (train / val / test) of the artists under `root`."""

    def __init__(self, root, split, loader, metadata, split_proportions=(8, 1, 1),
                 persist=True, cache=True, validate=True, min_songs=0, parallel=False,
                 valid_songs_file='valid_songs.csv', seed=None, song_order=None):
        self.root = root
        self.split = split
        self.cache = cache
        self.cache_data = {}
        self.loader = loader
        self.metadata = metadata
        self.valid_songs_file = valid_songs_file
        self.artists = []
        self._table = None

        valid = {}                    # artist -> ordered list of valid songs
        if validate and persist:
            for line in metadata.lines(valid_songs_file):
                artist, song = line.rstrip('\n').split(',', 1)
                songs = valid.setdefault(unquote(artist), [])
                song = unquote(song)
                if song not in songs:
                    songs.append(song)

        split_file = '%s.csv' % split
        if persist and metadata.exists(split_file):
            in_split = [line.rstrip('\n') for line in metadata.lines(split_file)]
        else:
            in_split = self._scan_and_split(valid, split, split_proportions, persist, validate,
                                            min_songs, seed)
        metadata.close()

        for artist in in_split:
            songs = valid[artist]
            if song_order is not None and artist in song_order:
                assert sorted(song_order[artist]) == sorted(songs), 'song_order must be a permutation'
                songs = song_order[artist]
            self.artists.append(ArtistDataset(artist, songs))

    def _scan_and_split(self, valid, split, proportions, persist, validate, min_songs, seed):
        loader, root = self.loader, self.root
        dirs = []
        for artist in sorted(os.listdir(root)):
            adir = os.path.join(root, artist)
            if os.path.isdir(adir) and any(loader.is_song(s) for s in os.listdir(adir)):
                dirs.append(artist)
        progress = ProgressLogger(len(dirs))
        kept, skipped = [], 0
        for index, artist in enumerate(dirs):
            songs = sorted(s for s in os.listdir(os.path.join(root, artist)) if loader.is_song(s))
            known = valid.setdefault(artist, [])
            if validate:
                progress.maybe_log(index)
                for song in songs:
                    if song in known:
                        continue
                    if loader.validate(os.path.join(root, artist, song)):
                        known.append(song)
                        if persist:
                            self.metadata.write(self.valid_songs_file,
                                                '%s,%s\n' % (quote(artist), quote(song)))
            else:
                valid[artist] = songs
            if len(valid[artist]) >= min_songs:
                kept.append(artist)
            else:
                skipped += 1
        if skipped:
            log.info("%s artists don't have K+K'=%s songs. Using %s artists" % (skipped, min_songs, len(kept)))
        total = float(sum(proportions))
        n_train = int(proportions[0] / total * len(kept))
        n_val = int(proportions[1] / total * len(kept))
        np.random.RandomState(seed).shuffle(kept)       # same shuffle call as dataset.py:170
        parts = {'train': kept[:n_train], 'val': kept[n_train:n_train + n_val], 'test': kept[n_train + n_val:]}
        if persist:
            for name in ('train', 'val', 'test'):
                # the reference appends '\n'.join(artists) with no trailing newline (dataset.py:171-174): two writers would glue
                # two artists into one bogus name.  One atomic write per split file; readers strip and skip blank lines
                self.metadata.write_whole('%s.csv' % name, '\n'.join(parts[name]) + ('\n' if parts[name] else ''))
        return parts.get(split, parts['test'])

    # -- access ------------------------------------------------------------------
    def load(self, artist, song):
        """(artist, song) -> int32 [max_len], cached in RAM (dataset.py:187-199)."""
        key = (artist, song)
        if self.cache and key in self.cache_data:
            return self.cache_data[key]
        row = self.loader.load(os.path.join(self.root, artist, song))
        self.cache_data[key] = row
        return row

    def token_table(self):
        """Packed view of the split: (table int32 [n_songs, max_len], offsets int64 [n_artists+1]);
        artist a's songs are rows offsets[a]..offsets[a+1] in `artist.songs` order."""
        if self._table is None:
            rows, offsets = [], [0]
            for artist in self.artists:
                rows.extend(self.load(artist.name, song) for song in artist.songs)
                offsets.append(len(rows))
            table = np.ascontiguousarray(np.stack(rows).astype(np.int32)) if rows \
                else np.zeros((0, self.loader.max_len), np.int32)
            self._table = (table, np.asarray(offsets, np.int64))
        return self._table

    def __len__(self):
        return len(self.artists)

    def __getitem__(self, index):
        return self.artists[index]
