import os, sys
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
cfg = dict(bench.CFG_B); N, K, Q = 5, 5, 4
eps = bench.synthetic_episodes(3, N, K, Q, cfg['max_len'], cfg['input_size'], 5)
m = FsmgModel(cfg); m.init_params(3)
for i in range(3):
    m.forward_backward(*eps[i])
    g = {k: m.get_grad(k) for k in m.param_shapes}
    print('pass', i, {k: float(np.abs(v).max()) for k, v in g.items()}, 'tail', m.debug_read('tail', 4))
    if i == 1:
        kk = g['kernel_0']; bad = np.argwhere(np.abs(kk) > 1e3)
        print('  big entries:', len(bad), bad[:5].tolist(), bad[-3:].tolist() if len(bad) else '')
    print('  loss', m.apply_update(1.0), m.stats())
d = m.debug_dims(); T, Hp = d['T'], d['Hp']; B = 45
h = m.debug_read('h0', (T + 1) * B * Hp).reshape(T + 1, B, Hp)
print('h index 0 block: max |.|', float(np.abs(h[0]).max()), 'nonzero', int((h[0] != 0).sum()), 'finite', bool(np.isfinite(h).all()))
nz = np.argwhere(h[0] != 0)
if len(nz): print('  nonzero rows', sorted(set(nz[:, 0].tolist()))[:20], 'cols', nz[:, 1].min(), nz[:, 1].max(), h[0][h[0] != 0][:6])
c = m.debug_read('c0', (T + 1) * B * Hp).reshape(T + 1, B, Hp)
print('c index 0 block: max', float(np.abs(c[0]).max()))
