"""The C++/OpenMP CPU restatement (oracle/cpu_ref.cpp, bench.py's second cpu_baseline variant) against the numpy oracle."""
import os

import numpy as np
import pytest

from conftest import small_config
from oracle import lstm_oracle as O


@pytest.mark.parametrize('L,mode', [(1, 'tf1_slices'), (2, 'dense')])
def test_cxx_restatement_matches_numpy_oracle(L, mode):
    from oracle import cpu_ref
    if not os.path.isfile(cpu_ref.LIB):
        cpu_ref.build()
    cfg = small_config(n_layers=L, hidden_size=24, embedding_size=12, input_size=61, max_len=9, max_grad_norm=0.3, n_decay=7)
    params = O.glorot_init(cfg, 4, np.float64)
    ref = cpu_ref.CpuRef(cfg, params, threads=2, clip_norm_mode=mode)
    opt = O.new_opt_state(params)
    eps = O.synthetic_episodes(4, 3, 2, 2, cfg['max_len'], cfg['input_size'], seed=6, realistic=True)
    assert abs(ref.eval(eps[0][1]) - O.eval_step(params, eps[0][1], cfg)) < 1e-5
    for sup, qry in eps:
        want = O.train_step(params, opt, sup, qry, cfg, clip_norm_mode=mode)
        assert abs(ref.train(sup, qry) - want) <= 1e-5 * abs(want)
    got = ref.get_params()
    for k, v in params.items():
        assert np.abs(got[k] - v).max() <= 2e-4 * max(np.abs(v).max(), 1e-6), k
