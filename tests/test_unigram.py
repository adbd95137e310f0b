"""UnigramModel against hand-computed answers that follow the reference's slices
(/root/reference/src/models/unigram_model.py:41-69: train on tokens[:, :-1] of support + query, eval on tokens[:, 1:] of
the query, most frequent word as the sample).  CPU: the numpy restatement of the three lines of arithmetic and the word
slices; GPU (-m gpu): the plugin on the device-resident histogram (fsmg_unigram_*) against both."""
import numpy as np
import pytest

from data.episode import Episode
from models.unigram_model import eval_words, host_unigram_nll, host_unigram_update, train_words

SUP = np.array([[[0, 1, 1, 4]]], np.int32)          # 1 artist x 1 song: train words = [0, 1, 1] (last token dropped)
QRY = np.array([[[2, 2, 3, 4]]], np.int32)          # train words = [2, 2, 3]; eval words = [2, 3, 4] (first dropped)


def test_word_slices_and_host_restatement_known_answers():
    ep = Episode(SUP, QRY)
    assert train_words(ep).tolist() == [0, 1, 1, 2, 2, 3] and eval_words(ep).tolist() == [2, 3, 4]
    counts = np.full(5, 1.0, np.float32)
    assert abs(host_unigram_nll(counts, eval_words(ep)) - np.log(5.0)) < 1e-6       # before any update: uniform 1/5
    host_unigram_update(counts, train_words(ep))
    assert counts.tolist() == [2, 3, 3, 2, 1]                                       # the 4s in the last column were never counted
    want_eval = -np.mean(np.log(np.array([3, 2, 1]) / 11.0))
    assert abs(host_unigram_nll(counts, eval_words(ep)) - want_eval) < 1e-6


@pytest.mark.gpu
def test_plugin_known_answers_on_the_device(tmp_path):
    from models.unigram_model import UnigramModel
    m = UnigramModel(dict(name='unigram', input_size=5, max_len=4))
    ep = Episode(SUP, QRY)
    assert abs(m.eval(ep) - np.log(5.0)) < 1e-6
    assert abs(m.train(ep) - np.log(5.0)) < 1e-6        # loss with the counts BEFORE the update
    assert m.engine.get_counts().tolist() == [2, 3, 3, 2, 1]
    want_eval = -np.mean(np.log(np.array([3, 2, 1]) / 11.0))
    assert abs(m.eval(ep) - want_eval) < 1e-6
    want_train = -np.mean(np.log(np.array([2, 3, 3, 3, 3, 2]) / 11.0))
    assert abs(m.train(ep) - want_train) < 1e-6
    assert m.sample(SUP[0], 3) == [1, 1, 1]             # argmax picks the first of the most frequent words
    m.save(str(tmp_path))
    m2 = UnigramModel(dict(name='unigram', input_size=5, max_len=4))
    m2.recover_or_init(str(tmp_path))
    assert m2.eval(ep) == m.eval(ep)


@pytest.mark.gpu
def test_device_histogram_equals_the_host_restatement_on_episode_sized_input():
    """cfg-B sized episodes (Zipf-distributed ids: heavy duplicates in the scatter_add): counts bit-equal, NLL within fp32 rounding"""
    import torch
    from fsmg.binding import FsmgError, FsmgUnigram
    V, T = 10000, 128
    rng = np.random.RandomState(5)
    u = FsmgUnigram(V)
    counts = np.full(V, 1.0, np.float32)
    for step in range(6):
        sup = np.minimum(rng.zipf(1.2, size=(5, 5, T)) - 1, V - 1).astype(np.int32)
        qry = rng.randint(0, V, size=(5, 4, T)).astype(np.int32)
        ep = Episode(sup, qry)
        w = train_words(ep)
        want = host_unigram_nll(counts, w)
        if step % 2:                                    # device-resident words
            d = torch.from_numpy(w).cuda()
            got = u.train((d.data_ptr(), w.size))
            torch.cuda.synchronize()
        else:
            got = u.train(w)
        assert abs(got - want) <= 2e-6 * abs(want), (step, got, want)
        host_unigram_update(counts, w)
        assert np.array_equal(u.get_counts(), counts)
        e = eval_words(ep)
        assert abs(u.nll(e) - host_unigram_nll(counts, e)) <= 2e-6 * abs(want)
    assert u.argmax() == int(np.argmax(counts))
    with pytest.raises(FsmgError, match='TOKEN_RANGE'):
        u.nll(np.array([0, V], np.int32))
    before = u.get_counts()
    with pytest.raises(FsmgError, match='TOKEN_RANGE'):
        u.train(np.array([3, -1], np.int32))
    assert np.array_equal(u.get_counts(), before)       # a rejected batch leaves the counts alone
    frac = before.copy(); frac[7] += 0.5
    with pytest.raises(FsmgError, match='whole numbers'):   # the device holds integers: a fractional checkpoint is refused, not rounded
        u.set_counts(frac)
    assert np.array_equal(u.get_counts(), before)
    assert abs(u.nll(np.array([3], np.int32)) - host_unigram_nll(counts, np.array([3]))) < 1e-5
