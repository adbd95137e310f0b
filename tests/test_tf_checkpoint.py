"""models/tf_checkpoint.py -- the tensor-bundle reader that lets a checkpoint written by the reference's tf.train.Saver
(/root/reference/src/models/tf_model.py:96-125) be taken over.  No TensorFlow-written file exists in the reference tree and
TensorFlow cannot be installed here, so the reader is pinned by known vectors of its primitives and by round trips through
this module's own writer (plain and snappy-compressed blocks, multi-block indexes)."""
import os

import numpy as np
import pytest

from conftest import small_config
from models import tf_checkpoint as TC
from oracle import lstm_oracle as O


def test_primitives_known_vectors():
    assert TC.crc32c(b'123456789') == 0xE3069283                       # the standard CRC-32C check value
    assert TC.crc32c(b'\x00' * 32) == 0x8A9136AA                        # RFC 3720 B.4: 32 bytes of zeros
    assert TC.masked_crc(b'') == 0xa282ead8                             # mask(0) = rotate(0) + 0xa282ead8 (leveldb crc32c.h)
    # snappy: "Wikipedia" as one literal; then a stream with 1-, 2-byte-offset copies and an overlapping copy
    assert TC.snappy_decompress(bytes([9, 8 << 2]) + b'Wikipedia') == b'Wikipedia'
    s = bytes([18]) + bytes([2 << 2]) + b'abc' + bytes([((6 - 4) << 2) | 1, 3]) + bytes([((9 - 1) << 2) | 2, 9, 0])
    assert TC.snappy_decompress(s) == b'abc' + b'abcabc' + b'abcabcabc'
    assert TC._varint(bytes([0xAC, 0x02]), 0) == (300, 2) and TC._put_varint(300) == bytes([0xAC, 0x02])


@pytest.mark.parametrize('compress,block_entries', [(False, 16), (True, 3)])
def test_bundle_round_trip_and_reference_variable_names(tmp_path, compress, block_entries):
    cfg = small_config(n_layers=2, hidden_size=12, embedding_size=6, input_size=40)
    params = O.glorot_init(cfg, 3, np.float32)
    scope = 'lstm_baseline'
    tf_names = {'embedding': scope + '/embedding', 'softmax_w': scope + '/softmax_w', 'softmax_b': scope + '/softmax_b'}
    for l in range(2):
        tf_names['kernel_%d' % l] = '%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % (scope, l)
        tf_names['bias_%d' % l] = '%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % (scope, l)
    tensors = {scope + '/Variable': np.array(4321, np.int32), scope + '/beta1_power': np.array(0.9 ** 4321, np.float32)}
    for k, v in params.items():
        tensors[tf_names[k]] = v
        tensors[tf_names[k] + '/Adam'] = (v * 0.5).astype(np.float32)
        tensors[tf_names[k] + '/Adam_1'] = (v * v).astype(np.float32)
    prefix = str(tmp_path / scope / (scope + '-4321'))
    TC.write_bundle(prefix, tensors, block_entries=block_entries, compress=compress)
    assert TC.latest_checkpoint(str(tmp_path / scope)) == prefix
    back = TC.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k in tensors:
        np.testing.assert_array_equal(back[k], tensors[k])
        assert back[k].dtype == tensors[k].dtype and back[k].shape == tensors[k].shape
    p, m, v, step = TC.map_variables(back, 2)
    assert step == 4321 and set(p) == set(params) == set(m) == set(v)
    for k in params:
        np.testing.assert_array_equal(p[k], params[k])
        np.testing.assert_array_equal(v[k], (params[k] * params[k]).astype(np.float32))
    # a flipped payload byte is caught by the per-tensor checksum, a flipped index byte by the block checksum
    data = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data, 'rb').read()); raw[10] ^= 1; open(data, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        TC.read_bundle(prefix)
    idx = bytearray(open(prefix + '.index', 'rb').read()); idx[20] ^= 1; open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum|corrupt|mismatch'):
        TC.read_index(prefix + '.index')


def test_pre_tf12_names_and_missing_state_file(tmp_path):
    t = {'m/embedding': np.zeros((3, 2), np.float32), 'm/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/weights': np.ones((4, 8), np.float32),
         'm/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/biases': np.ones(8, np.float32), 'm/Variable': np.array(7, np.int64)}
    TC.write_bundle(str(tmp_path / 'm-7'), t)
    TC.write_bundle(str(tmp_path / 'm-12'), t)
    os.remove(str(tmp_path / 'checkpoint'))
    assert TC.latest_checkpoint(str(tmp_path)) == str(tmp_path / 'm-12')
    p, _, _, step = TC.map_variables(TC.read_bundle(str(tmp_path / 'm-7')), 1)
    assert step == 7 and set(p) == {'embedding', 'kernel_0', 'bias_0'}
    assert TC.latest_checkpoint(str(tmp_path / 'nope')) is None


@pytest.mark.gpu
def test_plugin_takes_over_a_tensorflow_checkpoint(tmp_path):
    from gpu_utils import new_model
    from models.lstm_baseline import LSTMBaseline
    cfg = small_config(n_layers=2, hidden_size=24, embedding_size=12, input_size=61)
    params = O.glorot_init(cfg, 8, np.float32)
    scope = cfg['name']
    names = {'embedding': 'embedding', 'softmax_w': 'softmax_w', 'softmax_b': 'softmax_b'}
    for l in range(2):
        names['kernel_%d' % l] = 'rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % l
        names['bias_%d' % l] = 'rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % l
    tensors = {scope + '/Variable': np.array(55, np.int32)}
    for k, v in params.items():
        tensors['%s/%s' % (scope, names[k])] = v
        tensors['%s/%s/Adam' % (scope, names[k])] = (0.1 * v).astype(np.float32)
        tensors['%s/%s/Adam_1' % (scope, names[k])] = (v * v).astype(np.float32)
    TC.write_bundle(str(tmp_path / scope / (scope + '-55')), tensors, compress=True)
    m = LSTMBaseline(dict(cfg))
    m.recover_or_init(str(tmp_path))
    assert m.engine.step == 55
    for k, v in params.items():
        np.testing.assert_array_equal(m.engine.get_param(k), v)
        am, av = m.engine.get_opt_state(k)
        np.testing.assert_array_equal(av, (v * v).astype(np.float32))
    # and training continues from there exactly like a handle given the same state by hand
    ref = new_model(cfg, params=params)
    for k, v in params.items():
        ref.set_opt_state(k, (0.1 * v).astype(np.float32), (v * v).astype(np.float32))
    ref.step = 55
    (sup, qry), = O.synthetic_episodes(1, 2, 2, 1, cfg['max_len'], cfg['input_size'], seed=1)
    from data.episode import Episode
    assert m.train(Episode(sup, qry)) == ref.train_step(sup, qry)
