"""Helpers shared by the -m gpu tests (layout conversions of the debug buffers)."""
import numpy as np

from oracle import lstm_oracle as O


def new_model(cfg, params=None, **kw):
    from fsmg.binding import FsmgModel
    m = FsmgModel(cfg, **kw)
    if params is None:
        m.init_params(cfg.get('seed', 0))
    else:
        m.init_params(0)
        m.set_params({k: np.asarray(v, np.float32) for k, v in params.items()})
    return m


def f64_params(model):
    return {k: v.astype(np.float64) for k, v in model.get_params().items()}


def read_states(model, cfg, layer, B):
    """-> hs [T+1,B,H], cs [T+1,B,H], gates [T,4,B,H] from the padded / packed debug buffers."""
    d = model.debug_dims()
    T, Hp, H = d['T'], d['Hp'], cfg['hidden_size']
    hs = model.debug_read('h%d' % layer, (T + 1) * B * Hp).reshape(T + 1, B, Hp)
    cs = model.debug_read('c%d' % layer, (T + 1) * B * Hp).reshape(T + 1, B, Hp)
    g = model.debug_read('gates%d' % layer, T * B * 4 * Hp).reshape(T, B, Hp // 4, 4, 4)   # [.., unit block, gate, unit%4]
    gates = np.transpose(g, (0, 3, 1, 2, 4)).reshape(T, 4, B, Hp)
    assert np.all(hs[:, :, H:] == 0) and np.all(cs[:, :, H:] == 0), 'padded hidden units must stay exactly zero'
    return hs[:, :, :H], cs[:, :, :H], gates[:, :, :, :H]


def time_major(a, B, T):
    """oracle row order b*T+t -> device row order t*B+b"""
    return a.reshape(B, T, *a.shape[1:]).swapaxes(0, 1).reshape(B * T, *a.shape[1:])


def rel_max(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def oracle_step(params, sup, qry, cfg):
    X, Y = O.train_xy(sup, qry, cfg['input_size'])
    loss, cache = O.forward(params, X, Y, cfg)
    grads, aux = O.backward(params, cache, cfg)
    return float(loss), cache, grads, aux
