"""UnigramModel against hand-computed answers that follow the reference's slices
(/root/reference/src/models/unigram_model.py:41-69: train on tokens[:, :-1] of support + query, eval on tokens[:, 1:] of
the query, most frequent word as the sample)."""
import numpy as np

from data.episode import Episode
from models.unigram_model import UnigramModel


def test_known_answers(tmp_path):
    m = UnigramModel(dict(name='unigram', input_size=5, max_len=4))
    sup = np.array([[[0, 1, 1, 4]]], np.int32)          # 1 artist x 1 song: train words = [0, 1, 1] (last token dropped)
    qry = np.array([[[2, 2, 3, 4]]], np.int32)          # train words = [2, 2, 3]; eval words = [2, 3, 4] (first dropped)
    ep = Episode(sup, qry)
    # before any update: uniform 1/5
    assert abs(m.eval(ep) - np.log(5.0)) < 1e-6
    assert abs(m.train(ep) - np.log(5.0)) < 1e-6        # loss with the counts BEFORE the update
    # counts = 1 + [1, 2, 2, 1, 0] = [2, 3, 3, 2, 1], sum 11; the 4s in the last column were never counted
    want_eval = -np.mean(np.log(np.array([3, 2, 1]) / 11.0))
    assert abs(m.eval(ep) - want_eval) < 1e-6
    want_train = -np.mean(np.log(np.array([2, 3, 3, 3, 3, 2]) / 11.0))
    assert abs(m.train(ep) - want_train) < 1e-6
    assert m.sample(sup[0], 3) == [1, 1, 1]             # argmax picks the first of the most frequent words
    m.save(str(tmp_path))
    m2 = UnigramModel(dict(name='unigram', input_size=5, max_len=4))
    m2.recover_or_init(str(tmp_path))
    assert m2.eval(ep) == m.eval(ep)
