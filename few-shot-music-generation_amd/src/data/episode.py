"""Episode sampling: artist-grouped N-way / K-shot episodes.

Host mirror of /root/reference/src/data/episode.py.  Contract kept bit-for-bit
(pinned by tests/golden/g2_episodes.npz, captured from the reference):

  * one legacy `RandomState(seed)` stream per sampler (episode.py:53,152-156);
  * per episode it is consumed as `permutation(n_artists)` once (the first N are the
    episode's artists, == RandomState.choice(dataset, N, replace=False), episode.py:65)
    and then `permutation(n_songs)` once per chosen artist in order (episode.py:34-38);
    the first Q drawn songs are the QUERY set, the remaining K the support set
    (episode.py:39-40);
  * Episode.support int32 [N,K,T], Episode.query int32 [N,Q,T], C-contiguous.

MI355X-first differences: the split is one packed int32 token table and an episode
is a single index gather from it (`episode_indices` + `gather`), so the same table
can live in HBM and whole pools of episodes can be materialised at once
(`get_episodes`); `ShardedEpisodeSampler` deals consecutive episodes of ONE stream
round-robin to the ranks of an episode-parallel job.
"""
import os

import numpy as np
import yaml
from numpy.random import RandomState

from data.dataset import Dataset, Metadata
from data.lyrics_loader import LyricsLoader
from data.midi_loader import MIDILoader


class Episode(object):
    def __init__(self, support, query):
        self.support = support
        self.query = query


class SQSampler(object):
    """Draws K+Q distinct songs of one artist; query songs come first."""

    def __init__(self, support_size, query_size, random):
        self.support_size = support_size
        self.query_size = query_size
        self.random = random

    def sample_indices(self, n_songs):
        picked = self.random.permutation(n_songs)[:self.support_size + self.query_size]
        return picked[:self.query_size], picked[self.query_size:]

    def sample(self, artist):
        query, support = self.sample_indices(len(artist))
        return [artist[i] for i in query], [artist[i] for i in support]


class EpisodeSampler(object):
    def __init__(self, dataset, batch_size, support_size, query_size, max_len,
                 dtype=np.int32, seed=None):
        self.dataset = dataset
        self.batch_size = batch_size
        self.support_size = support_size
        self.query_size = query_size
        self.max_len = max_len
        self.dtype = dtype
        self.random = get_random(seed)
        self.sq_sampler = SQSampler(support_size, query_size, self.random)

    def __len__(self):
        return len(self.dataset)

    def __repr__(self):
        return 'EpisodeSampler(%r, %r)' % (getattr(self.dataset, 'root', None), getattr(self.dataset, 'split', None))

    def episode_indices(self):
        """Row indices into the packed token table: (support [N,K], query [N,Q])."""
        n_artists = len(self.dataset)
        if n_artists < self.batch_size:
            raise ValueError('Cannot take a larger sample than population when replace is False')
        _, offsets = self.dataset.token_table()
        artists = self.random.permutation(n_artists)[:self.batch_size]
        support = np.empty((self.batch_size, self.support_size), np.int64)
        query = np.empty((self.batch_size, self.query_size), np.int64)
        for b, a in enumerate(artists):
            n_songs = int(offsets[a + 1] - offsets[a])
            if n_songs < self.support_size + self.query_size:
                raise ValueError('Cannot take a larger sample than population when replace is False')
            q, s = self.sq_sampler.sample_indices(n_songs)
            support[b] = offsets[a] + s
            query[b] = offsets[a] + q
        return support, query

    def gather(self, support_idx, query_idx):
        table, _ = self.dataset.token_table()
        return Episode(np.ascontiguousarray(table[support_idx], dtype=self.dtype),
                       np.ascontiguousarray(table[query_idx], dtype=self.dtype))

    def get_episode(self):
        return self.gather(*self.episode_indices())

    def next_indices(self):
        """the next episode of the stream as (support [N,K], query [N,Q]) row indices into token_table() -- same RNG
        consumption as get_episode()"""
        return self.episode_indices()

    def token_table(self):
        return self.dataset.token_table()[0]

    def get_episodes(self, n):
        """The next n episodes of the stream (same as n get_episode() calls)."""
        return [self.get_episode() for _ in range(n)]

    def get_num_unique_words(self):
        return self.dataset.loader.get_num_tokens()

    def detokenize(self, numpy_data):
        return self.dataset.loader.detokenize(numpy_data)


class ShardedEpisodeSampler(object):
    """Rank r of R sees episodes r, r+R, r+2R, ... of the wrapped sampler's stream,
    i.e. outer step s trains on episodes s*R .. s*R+R-1, one per rank -- the same
    episode sequence a 1-GPU run with R-episode accumulation would see."""

    def __init__(self, sampler, rank, world_size):
        assert 0 <= rank < world_size
        self.sampler, self.rank, self.world_size = sampler, rank, world_size

    def next_indices(self):
        mine = None
        for r in range(self.world_size):
            idx = self.sampler.episode_indices()
            if r == self.rank:
                mine = idx
        return mine

    def get_episode(self):
        return self.sampler.gather(*self.next_indices())

    def __getattr__(self, name):
        return getattr(self.sampler, name)


_REQUIRED = ('dataset_path', 'query_size', 'support_size', 'batch_size', 'max_len', 'dataset', 'split')


def load_sampler_from_config(config):
    """YAML path / dict / stream -> EpisodeSampler (episode.py:82-149).  Keys: _REQUIRED plus the
    optional train_/val_/test_proportion, persist, cache, validate, seed, dataset_seed."""
    if isinstance(config, str):
        with open(config, 'r') as f:
            config = yaml.safe_load(f)              # Q1: yaml.load without Loader fails on PyYAML >= 6
    elif not isinstance(config, dict):
        config = yaml.safe_load(config)
    for key in _REQUIRED:
        if key not in config:
            raise RuntimeError('required config key "%s" not found' % key)
    root = config['dataset_path']
    if not os.path.isdir(root):
        raise RuntimeError('required data directory %s does not exist' % root)
    metadata = Metadata(root, 'few_shot_metadata_%s_%s' % (config['dataset'], config['max_len']))
    if config['dataset'] == 'lyrics':
        loader = LyricsLoader(config['max_len'], metadata=metadata)
    elif config['dataset'] == 'midi':
        loader = MIDILoader(config['max_len'])
    else:
        raise RuntimeError('unknown dataset "%s"' % config['dataset'])
    dataset = Dataset(
        root, config['split'], loader, metadata,
        split_proportions=(config.get('train_proportion', 8), config.get('val_proportion', 1),
                           config.get('test_proportion', 1)),
        cache=config.get('cache', True), persist=config.get('persist', True),
        validate=config.get('validate', True),
        min_songs=config['support_size'] + config['query_size'],
        parallel=False, seed=config.get('dataset_seed', 0),
        song_order=config.get('_song_order'))
    return EpisodeSampler(dataset, config['batch_size'], config['support_size'], config['query_size'],
                          config['max_len'], seed=config.get('seed', None))


def get_random(seed):
    """Legacy MT19937 RandomState(seed), or the global numpy stream when seed is None."""
    return RandomState(seed) if seed is not None else np.random
