#!/usr/bin/env python3
"""Where does the XCD-partitioned schedule differ from the serial one?  One full-size cfg-B pass per handle (FSMG_XCD_BX3=1 both),
then h, logits, dlogits and every gradient compared element by element; mismatches are reported per 256 x 256 projection tile /
per time step so that the pattern names the culprit.  python tools/xov_check.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
from conftest import small_config
from gpu_utils import new_model
from oracle import lstm_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = small_config(input_size=10000, max_len=128, embedding_size=250, hidden_size=512, n_layers=1)
N, K, Q = 5, 5, 4
B, T = N * (K + Q), cfg['max_len']
eps = O.synthetic_episodes(steps, N, K, Q, T, cfg['input_size'], seed=29)
os.environ['FSMG_XCD_BX3'] = '1'
out = {}
for name, env in (('serial', {'FSMG_XCD_OVERLAP': '0'}), ('xov', {'FSMG_XCD_OVERLAP': '1', 'FSMG_XOV_PARTS': os.environ.get('PARTS', '1')})):
    os.environ.update(env)
    m = new_model(cfg)
    m.debug_set('inplace_dlogits', 0)       # the comparison reads the logits: keep them (round 5: dlogits overwrite them by default)
    rec = []
    for s, q in eps:
        if os.environ.get('FUSED', '1') == '1':
            loss = m.train_step(s, q)
        else:
            m.forward_backward(s, q)
        d = m.debug_dims()
        rec.append({'h': m.debug_read('h0', (T + 1) * B * d['Hp']).reshape(T + 1, B, d['Hp']),
                    'logits': m.debug_read('logits', B * T * d['V1p']).reshape(T * B, d['V1p']),
                    'ce': m.debug_read('ce', B * T), 'dh': m.debug_read('dh', B * T * d['Hp']),
                    'tail': m.debug_read('tail', 16), 'stats': m.stats()})
        if os.environ.get('FUSED', '1') != '1':
            m.apply_update(1.0)
    out[name] = rec
for i in range(steps):
    a, b = out['serial'][i], out['xov'][i]
    print('step %d: loss serial %r xov %r; stats %r' % (i, a['tail'][1], b['tail'][1], b['stats']))
    for k_ in ('ce', 'dh'):
        dd = (a[k_] != b[k_]) | np.isnan(b[k_])
        print('   %s: %d differing, NaN %d (serial NaN %d)' % (k_, dd.sum(), np.isnan(b[k_]).sum(), np.isnan(a[k_]).sum()))
    hd = (a['h'] != b['h']) | np.isnan(b['h'])
    print('   h: %d differing elements, NaN in xov %d; time indices with differences: %r' % (hd.sum(), np.isnan(b['h']).sum(), np.unique(np.nonzero(hd)[0])[:20]))
    ld = (a['logits'] != b['logits']) | np.isnan(b['logits'])
    print('   logits: %d differing elements, NaN %d' % (ld.sum(), np.isnan(b['logits']).sum()))
    if ld.any():
        rows, cols = np.nonzero(ld)
        tiles = sorted(set(zip((rows // 256).tolist(), (cols // 256).tolist())))
        print('   tiles (row tile, col tile) with differences: %d of %d: %r' % (len(tiles), ((T * B + 255) // 256) * ((a['logits'].shape[1] + 255) // 256), tiles[:40]))
        for (tr, tc) in tiles[:6]:
            sa_, sb_ = a['logits'][256 * tr:256 * tr + 256, 256 * tc:256 * tc + 256], b['logits'][256 * tr:256 * tr + 256, 256 * tc:256 * tc + 256]
            per = []
            for t_ in range(256 * tr // B, (256 * tr + 255) // B + 1):
                r0, r1 = max(t_ * B, 256 * tr) - 256 * tr, min((t_ + 1) * B, 256 * tr + 256) - 256 * tr
                blk_a, blk_b = sa_[r0:r1], sb_[r0:r1]
                per.append('t%d:%s' % (t_, 'nan' if np.isnan(blk_b).all() else 'NAN+' if np.isnan(blk_b).any() else 'eq' if np.array_equal(blk_a, blk_b) else 'diff(max %.2g)' % np.abs(blk_a - blk_b).max()))
            print('      tile (%d, %d): %s' % (tr, tc, ' '.join(per)))
        rt = rows // 256
        for t_ in np.unique(rt)[:6]:
            rr = rows[rt == t_]
            print('      row tile %d: rows %d..%d differ (time steps %d..%d), %d elements; rows of the tile: %d..%d = steps %d..%d' % (
                t_, rr.min(), rr.max(), rr.min() // B, rr.max() // B, (rt == t_).sum(), 256 * t_, 256 * t_ + 255, 256 * t_ // B, (256 * t_ + 255) // B))
