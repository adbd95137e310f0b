#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the authoring container (needs /root/reference); the GPU box and the
tests only ever see the outputs (data, no reference code).  Recipe: SURVEY.md
Appendix C.  Run as:

    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/make_golden.py

Outputs
  g1_lyrics/                     synthetic mini lyrics dataset (24 artists): <song>.txt, the
                                 reference-written <song>.txt.32.npy sidecars and
                                 few_shot_metadata_lyrics_32/{word_ids,valid_songs,train,val,test}.csv
  g2_episodes.npz                first 8 episodes per split from the reference EpisodeSampler(seed=1234)
  g2_song_order.json             per-artist song order the reference realised (list(set) order, Q3)
  g3_xy.npz                      reference convert_tokens_to_input_and_target on the G2 episodes
  g4_config.json                 lyrics.yaml + 5shot.yaml + lstm_baseline.yaml merged as train.py:49-51 does
"""
import json
import os
import shutil
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'
T, K, Q = 32, 5, 4
N_EPISODES = 8


def build_dataset(root):
    rng = np.random.RandomState(42)
    if os.path.isdir(root):
        shutil.rmtree(root)
    os.makedirs(root)
    vocab = ['w%d' % i for i in range(300)]
    for a in range(24):
        adir = os.path.join(root, 'artist_%02d' % a)
        os.makedirs(adir)
        n_songs = 9 + (a * 7) % 4                     # 9..12, non-uniform (Q2)
        for s in range(n_songs):
            length = int(rng.randint(6, 48))          # some shorter than T (zero padded), some truncated
            words = rng.choice(vocab, size=length, p=None)
            with open(os.path.join(adir, 'song_%02d.txt' % s), 'w') as f:
                f.write(' '.join(words) + '\n')


def main():
    assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0 (Q3)'
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    nltk = types.ModuleType('nltk')
    nltk.word_tokenize = str.split                     # only evaluated as a default argument
    sys.modules['nltk'] = nltk
    sys.modules['pretty_midi'] = types.ModuleType('pretty_midi')

    from data import episode as ref_episode            # noqa: E402  (reference code)
    from data.dataset import Dataset                   # noqa: E402
    from models.base_model import convert_tokens_to_input_and_target  # noqa: E402

    def _as_array(self, dtype=None, copy=None):        # numpy>=1.24 shim (Q2)
        arr = np.empty(len(self.artists), dtype=object)
        for i, a in enumerate(self.artists):
            arr[i] = a
        return arr
    Dataset.__array__ = _as_array

    root = os.path.join(HERE, 'g1_lyrics')
    build_dataset(root)

    episodes, song_order, xy = {}, {}, {}
    for n_way, tag in ((2, 'n2'), (5, 'n5')):
        for split in ('train', 'val', 'test'):
            if n_way == 5 and split != 'train':
                continue                                # val/test splits hold < 5 artists
            cfg = dict(dataset='lyrics', dataset_path=os.path.abspath(root), max_len=T, query_size=Q,
                       support_size=K, batch_size=n_way, seed=1234, split=split)
            sampler = ref_episode.load_sampler_from_config(cfg)
            vocab = sampler.get_num_unique_words()
            song_order[split] = {a.name: list(a.songs) for a in sampler.dataset.artists}
            for e in range(N_EPISODES):
                ep = sampler.get_episode()
                key = '%s_%s_%d' % (tag, split, e)
                episodes[key + '_support'] = ep.support
                episodes[key + '_query'] = ep.query
                if tag == 'n2' and e < 2:
                    for nm, arr in (('support', ep.support), ('query', ep.query)):
                        x, y = convert_tokens_to_input_and_target(arr, vocab)
                        x0, y0 = convert_tokens_to_input_and_target(arr, None)
                        xy['%s_%s_x' % (key, nm)] = np.asarray(x)
                        xy['%s_%s_y' % (key, nm)] = np.asarray(y)
                        xy['%s_%s_x_nostart' % (key, nm)] = np.asarray(x0)
                        xy['%s_%s_y_nostart' % (key, nm)] = np.asarray(y0)
    episodes['vocab'] = np.int64(vocab)
    np.savez_compressed(os.path.join(HERE, 'g2_episodes.npz'), **episodes)
    np.savez_compressed(os.path.join(HERE, 'g3_xy.npz'), **xy)
    with open(os.path.join(HERE, 'g2_song_order.json'), 'w') as f:
        json.dump(song_order, f, indent=0, sort_keys=True)

    cfgdir = '/root/reference/src/config'
    merged = yaml.safe_load(open(os.path.join(cfgdir, 'lyrics.yaml')))
    merged.update(yaml.safe_load(open(os.path.join(cfgdir, '5shot.yaml'))))
    merged.update(yaml.safe_load(open(os.path.join(cfgdir, 'lstm_baseline.yaml'))))
    typed = {k: [type(v).__name__, v] for k, v in merged.items()}
    with open(os.path.join(HERE, 'g4_config.json'), 'w') as f:
        json.dump(typed, f, indent=1, sort_keys=True)
    print('vocab', vocab, 'episodes', len(episodes) - 1, 'xy', len(xy))


if __name__ == '__main__':
    main()
