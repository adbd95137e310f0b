import os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
os.environ.setdefault('FSMG_XOV_HEAD', '16')
from test_gpu_parity import FULL, small_config, new_model, O
over, N, K, Q = FULL['cfg-B']
cfg = small_config(**over)
eps = O.synthetic_episodes(3, N, K, Q, cfg['max_len'], cfg['input_size'], seed=23)
out = []
for x in ('0', '1'):
    os.environ['FSMG_XCD_OVERLAP'] = x
    m = new_model(cfg)
    losses = [m.train_step(s_, q_) for s_, q_ in eps]
    m.forward_backward(*eps[0])
    grads = {k: m.get_grad(k) for k in m.param_shapes}
    out.append((losses, grads))
    print(x, losses, flush=True)
print('loss equal', out[0][0] == out[1][0])
for k in out[0][1]:
    d = np.abs(out[0][1][k] - out[1][1][k]).max()
    print(k, 'maxdiff', d, 'ref max', np.abs(out[0][1][k]).max())
