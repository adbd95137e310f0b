"""train.train end to end: flags, YAML merge, loop cadence, log lines, checkpoints, sample dump.
CPU: with a fake plugin.  GPU: with the real LSTMBaseline plugin on the golden lyrics fixture."""
import os
import shutil
import sys

import pytest
import yaml

import train.train as T

K, Q, MAXLEN = 5, 4, 32


def _write_configs(tmp_path, golden_dir, model_cfg):
    root = tmp_path / 'g1_lyrics'
    shutil.copytree(os.path.join(golden_dir, 'g1_lyrics'), root)
    paths = {}
    docs = {
        'data': dict(dataset='lyrics', dataset_path=str(root), splits=['train', 'val', 'test'], max_len=MAXLEN),
        'task': dict(query_size=Q, support_size=K, seed=1234),
        'model': model_cfg,
    }
    for name, doc in docs.items():
        paths[name] = str(tmp_path / (name + '.yaml'))
        with open(paths[name], 'w') as f:
            yaml.safe_dump(doc, f)
    return paths


LOOP = dict(n_train=4, print_every_n=2, val_every_n=2.0, n_val=3, n_test=2, n_samples=2, batch_size=2)


def test_main_with_fake_plugin(tmp_path, golden_dir, capsys):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(LOOP, name='fake', model_module_name='fake_model', model_class_name='FakeModel')
    p = _write_configs(tmp_path, golden_dir, cfg)
    ck = str(tmp_path / 'ck')
    T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', ck])
    out = capsys.readouterr().out
    assert 'Num unique words: 300' in out
    assert 'Iter: 0, val-nll: 2.500e+00' in out
    assert 'Iter: 2, val-nll: 2.500e+00' in out and 'Iter: 4, val-nll: 2.500e+00' in out
    assert 'Iter: 2, loss: 7.500e+00' in out            # mean(10/1, 10/2)
    assert 'Iter: 4, loss: 2.917e+00' in out            # mean(10/3, 10/4)
    for line in ('Train Avg NLL: 2.500e+00', 'Validation Avg NLL: 2.500e+00', 'Test Avg NLL: 2.500e+00'):
        assert line in out
    from fake_model import FakeModel
    kinds = [c[0] for c in FakeModel.calls]
    assert kinds.count('train') == 4 and kinds.count('save') == 2 and kinds.count('eval') == 3 * 3 + 3 * 2
    assert FakeModel.calls[0] == ('init', 300) and FakeModel.calls[1] == ('recover_or_init', '')
    assert ('train', (2, K, MAXLEN), (2, Q, MAXLEN)) in FakeModel.calls
    for i in range(2):
        d = os.path.join(ck, 'samples', 'sample_%d' % i)
        assert sorted(os.listdir(d)) == ['model_sample.txt'] + ['support_%d.txt' % j for j in range(K)]
        assert open(os.path.join(d, 'model_sample.txt')).read().split()[0].startswith('w')
    # re-running into the same checkpt_dir must not crash at the sampling phase (Q10)
    T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', ck])


def test_config_merge_order_and_missing_vocab(tmp_path, golden_dir):
    cfg = dict(LOOP, name='fake', model_module_name='fake_model', model_class_name='FakeModel', max_len=MAXLEN, seed=7)
    p = _write_configs(tmp_path, golden_dir, cfg)
    args = T.build_parser().parse_args(['--data', p['data'], '--task', p['task'], '--model', p['model']])
    merged = T.load_config(args)
    assert merged['seed'] == 7                       # model yaml wins over task yaml
    assert merged['checkpt_dir'] == '' and os.path.isabs(merged['dataset_path'])
    assert vars(T.build_parser().parse_args([])) == dict(data='', model='', task='', checkpt_dir='', init_dir='')


def test_episode_size_for_the_kernel_choice_comes_from_the_merged_yamls(tmp_path, golden_dir):
    """fsmg_config.max_sequences (which recurrent kernel family a handle gets, include/fsmg.h) = N x (K + Q) of the merged YAMLs
    unless the model YAML names it; the shipped configs give cfg-B's 45 rows."""
    from models.hip_model import episode_sequences
    cfg = dict(LOOP, name='fake', model_module_name='fake_model', model_class_name='FakeModel', max_len=MAXLEN, batch_size=20)
    p = _write_configs(tmp_path, golden_dir, cfg)
    merged = T.load_config(T.build_parser().parse_args(['--data', p['data'], '--task', p['task'], '--model', p['model']]))
    assert episode_sequences(merged) == 20 * (K + Q)
    assert episode_sequences(dict(merged, max_sequences=64)) == 64
    assert episode_sequences({'batch_size': 5}) == 0                   # incomplete: the library's default
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'few-shot-music-generation_amd', 'src', 'config')
    shipped = {}
    for name in ('5shot.yaml', 'lstm_baseline.yaml'):
        with open(os.path.join(src, name)) as f:
            shipped.update(yaml.safe_load(f))
    assert episode_sequences(shipped) == 45


@pytest.mark.gpu
def test_unigram_plugin_stays_selectable(tmp_path, golden_dir, capsys):
    cfg = dict(LOOP, name='unigram_model', model_module_name='models.unigram_model', model_class_name='UnigramModel')
    cfg['batch_size'] = 1
    p = _write_configs(tmp_path, golden_dir, cfg)
    T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', str(tmp_path / 'u')])
    out = capsys.readouterr().out
    assert 'Test Avg NLL' in out and 'Iter: 4, loss:' in out


@pytest.mark.gpu
def test_main_with_lstm_baseline_on_gpu(tmp_path, golden_dir, capsys):
    cfg = dict(LOOP, name='lstm_baseline', model_module_name='models.lstm_baseline', model_class_name='LSTMBaseline',
               n_train=6, print_every_n=3, val_every_n=3.0, n_decay=10000, lr=5e-3, max_grad_norm=5,
               embedding_size=250, hidden_size=200, n_layers=1)
    p = _write_configs(tmp_path, golden_dir, cfg)
    ck = str(tmp_path / 'ck')
    T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', ck])
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if l.startswith('Iter: ') or 'Avg NLL' in l]
    assert lines[0].startswith('Iter: 0, val-nll: 5.7')          # ~ln(301) = 5.707 for untrained weights
    assert any(l.startswith('Iter: 6, loss: ') for l in lines) and any(l.startswith('Test Avg NLL') for l in lines)
    first = float(lines[0].split('val-nll: ')[1]); last = float([l for l in lines if 'val-nll' in l][-1].split('val-nll: ')[1])
    assert last < first                                            # six Adam steps already lower the val NLL
    assert os.path.isfile(os.path.join(ck, 'lstm_baseline', 'lstm_baseline-6.npz'))
    assert os.path.isfile(os.path.join(ck, 'samples', 'sample_1', 'model_sample.txt'))
    # resume from the checkpoint directory
    T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', str(tmp_path / 'ck2'),
            '--init_dir', ck])
    assert 'recovering lstm_baseline from' in capsys.readouterr().out
    # the run above took the fast path (episodes as indices into the device-resident table, losses read per log line);
    # the reference's calling convention (tokens + a loss read back per step) prints exactly the same lines
    os.environ['FSMG_TRAIN_SYNC'] = '1'
    try:
        T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', str(tmp_path / 'ck3')])
    finally:
        del os.environ['FSMG_TRAIN_SYNC']
    sync_lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith('Iter: ') or 'Avg NLL' in l]
    assert sync_lines == lines


@pytest.mark.gpu
def test_test_seed_entry_matches_oracle_trajectory(tmp_path, golden_dir, capsys):
    """train.test_seed (reference src/train/test_seed.py): the 10th train loss on the golden lyrics fixture equals
    the fp64 oracle's, started from the same parameters (the plugin's seeded init read back through the ABI)."""
    import numpy as np
    import train.test_seed as TS
    from data.episode import load_sampler_from_config
    from models.lstm_baseline import LSTMBaseline
    from oracle import lstm_oracle as O
    cfg = dict(LOOP, name='lstm_baseline', model_module_name='models.lstm_baseline', model_class_name='LSTMBaseline',
               n_decay=10000, lr=5e-3, max_grad_norm=5, embedding_size=64, hidden_size=48, n_layers=1, batch_size=5)
    p = _write_configs(tmp_path, golden_dir, cfg)
    # oracle side: same sampler stream, same initial parameters
    full = {}
    for k in ('data', 'task', 'model'):
        full.update(yaml.safe_load(open(p[k])))
    full['split'] = 'train'
    sampler = load_sampler_from_config(dict(full))
    full['input_size'] = sampler.get_num_unique_words()
    probe = LSTMBaseline(dict(full)); probe.recover_or_init('')
    params = {k: v.astype(np.float64) for k, v in probe.engine.get_params().items()}
    opt = O.new_opt_state(params)
    want = None
    for _ in range(TS.N_UPDATES):
        ep = sampler.get_episode()
        want = O.train_step(params, opt, ep.support, ep.query, full)
    assert TS.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--expect', '%.9f' % want]) == 0
    assert 'loss after 10 updates' in capsys.readouterr().out
    assert TS.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--expect', '%.9f' % (want + 0.5)]) == 1


@pytest.mark.gpu
def test_main_with_the_maml_plugin_trains_with_the_inner_loop(tmp_path, golden_dir, capsys):
    """`train.train --model maml_lstm.yaml` must run MAMLLSTM's step on BOTH calling conventions.  (Round 2's fast path called
    the inherited baseline train_indexed: inner loop skipped, training and evaluation silently disagreeing -- ADVICE r02.)
    The window means printed by the fast path equal the fp64 oracle's maml_step on the same sampler stream."""
    import numpy as np
    from data.episode import load_sampler_from_config
    from models.maml_lstm import MAMLLSTM
    from oracle import lstm_oracle as O
    cfg = dict(LOOP, name='maml_lstm', model_module_name='models.maml_lstm', model_class_name='MAMLLSTM',
               n_train=4, print_every_n=2, val_every_n=4.0, n_val=2, n_test=2, n_samples=1, n_decay=10000, lr=5e-3, max_grad_norm=5,
               embedding_size=32, hidden_size=40, n_layers=1, inner_steps=1, inner_lr=0.3)   # batch_size 2: the golden val split has two artists
    p = _write_configs(tmp_path, golden_dir, cfg)
    full = {}
    for k in ('data', 'task', 'model'):
        full.update(yaml.safe_load(open(p[k])))
    full['split'] = 'train'
    sampler = load_sampler_from_config(dict(full))
    full['input_size'] = sampler.get_num_unique_words()
    probe = MAMLLSTM(dict(full)); probe.recover_or_init('')
    params = {k: v.astype(np.float64) for k, v in probe.engine.get_params().items()}
    opt = O.new_opt_state(params)
    want = []
    for _ in range(4):
        ep = sampler.get_episode()
        want.append(O.maml_step(params, opt, ep.support, ep.query, full, 1, 0.3))
    # what the bug printed: plain baseline steps on the same stream (all rows at theta, no adaptation)
    sampler_b = load_sampler_from_config(dict(full))
    params_b = {k: v.astype(np.float64) for k, v in probe.engine.get_params().items()}
    opt_b = O.new_opt_state(params_b)
    plain = []
    for _ in range(2):
        ep = sampler_b.get_episode()
        plain.append(O.train_step(params_b, opt_b, ep.support, ep.query, full))
    out = {}
    for sync in ('0', '1'):
        os.environ['FSMG_TRAIN_SYNC'] = sync
        try:
            T.main(['--data', p['data'], '--task', p['task'], '--model', p['model'], '--checkpt_dir', str(tmp_path / ('ck' + sync))])
        finally:
            del os.environ['FSMG_TRAIN_SYNC']
        out[sync] = [l for l in capsys.readouterr().out.splitlines() if l.startswith('Iter: ') or 'Avg NLL' in l]
    assert out['0'] == out['1']                                    # fast path == the reference's calling convention
    loss_lines = [l for l in out['0'] if ', loss: ' in l]
    got = [float(l.split('loss: ')[1]) for l in loss_lines]
    assert len(got) == 2
    for g, w in zip(got, (np.mean(want[:2]), np.mean(want[2:]))):
        assert abs(g - w) <= 2e-3 * abs(w), (got, want)           # printed with 4 significant digits
    assert abs(np.mean(plain) - np.mean(want[:2])) > 5e-3 * abs(np.mean(want[:2]))      # the check has teeth
