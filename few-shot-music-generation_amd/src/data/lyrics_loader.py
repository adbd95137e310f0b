"""Lyrics loader: word tokens with a first-seen-order vocabulary.

Behaviour of the reference `LyricsLoader` (/root/reference/src/data/lyrics_loader.py:17-95):
ids are handed out in first-seen order and appended to `word_ids.csv`
(`<id>,<word>` lines) in the metadata directory; an existing file bootstraps the
vocabulary.  The default tokenizer is NLTK's `word_tokenize` when NLTK is
installed; it is looked up lazily so pre-tokenised datasets (the `.npy` sidecars
of base_loader) work without NLTK.
"""
import codecs
import logging
import string

import numpy as np

from data.base_loader import Loader

log = logging.getLogger('few-shot')


def _default_tokenizer(text):
    try:
        import nltk
    except ImportError:
        raise RuntimeError('raw lyrics need NLTK (word_tokenize); provide pre-tokenised '
                           '<song>.txt.<max_len>.npy sidecars or pass tokenizer=...')
    return nltk.word_tokenize(text)


class LyricsLoader(Loader):
    def __init__(self, max_len, metadata, tokenizer=None, persist=True, dtype=np.int32):
        super(LyricsLoader, self).__init__(max_len, dtype=dtype)
        self.tokenizer = tokenizer or _default_tokenizer
        self.metadata = metadata
        self.word_to_id = {}
        self.id_to_word = {}
        self.highest_word_id = -1
        if persist:
            log.info('Loading lyrics metadata...')
            for line in self.metadata.lines('word_ids.csv'):
                wid, word = line.rstrip('\n').split(',', 1)
                wid = int(wid)
                self.word_to_id[word] = wid
                self.id_to_word[wid] = word
                self.highest_word_id = max(self.highest_word_id, wid)

    def is_song(self, filepath):
        return filepath.endswith('.txt')

    def read(self, filepath):
        with codecs.open(filepath, 'r', errors='ignore') as f:
            return f.read()

    def get_num_tokens(self):
        return self.highest_word_id + 1

    def tokenize(self, raw_lyrics):
        ids = []
        for word in self.tokenizer(raw_lyrics):
            wid = self.word_to_id.get(word)
            if wid is None:
                self.highest_word_id += 1
                wid = self.highest_word_id
                self.word_to_id[word] = wid
                self.id_to_word[wid] = word
                if self.persist:
                    self.metadata.write('word_ids.csv', '%s,%s\n' % (wid, word))
            ids.append(wid)
        return ids

    def detokenize(self, numpy_data):
        """ids -> text; punctuation and clitics attach to the previous word
        (lyrics_loader.py:85-95)."""
        pieces = []
        for token in numpy_data:
            word = self.id_to_word[int(token)]
            glue = word == "n't" or word in string.punctuation or word.startswith("'")
            pieces.append(word if glue else ' ' + word)
        return ''.join(pieces).strip()
