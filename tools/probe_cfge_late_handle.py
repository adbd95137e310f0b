import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np, torch, bench
from fsmg.dist import EpisodeParallel
from models.lstm_baseline import LSTMBaseline
from models.maml_lstm import MAMLLSTM

def leg(name, steps=20, warmup=5, pool_n=32, keep=None):
    base, N, K, Q = bench.OTHER[name] if name != 'cfg-B' else (dict(bench.CFG_B), 5, 5, 4)
    cfg = dict(base, device=0); B, T = N * (K + Q), cfg['max_len']
    maml = (cfg['inner_steps'], cfg['inner_lr']) if name == 'cfg-E' else None
    pool = bench.synthetic_episodes(pool_n, N, K, Q, T, cfg['input_size'], seed=4321)
    d_sup = torch.from_numpy(np.stack([s for s, _ in pool])).cuda(); d_qry = torch.from_numpy(np.stack([q for _, q in pool])).cuda()
    ss, qs = d_sup[0].numel() * 4, d_qry[0].numel() * 4
    m = (MAMLLSTM if maml else LSTMBaseline)(dict(cfg, max_sequences=B)); m.recover_or_init('')
    par = EpisodeParallel(m); kw = dict(maml=maml) if maml else {}
    def step(i):
        e = i % len(pool); par.train_step(d_sup.data_ptr() + e * ss, d_qry.data_ptr() + e * qs, want_loss=False, shape=(N, K, Q), **kw)
    for i in range(warmup): step(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(4):
        t0 = time.perf_counter()
        for i in range(steps // 4): step(warmup + i)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0) / (steps // 4))
    print('%s (%s): ms/step per quarter %s' % (name, sys.argv[1], ' '.join('%.3f' % t for t in ts)), flush=True)
    if keep is not None: keep.append((m, par, d_sup, d_qry))

mode = sys.argv[1]
alive = []
if mode == 'alone': leg('cfg-E')
elif mode == 'after_B_alive': leg('cfg-B', keep=alive); leg('cfg-E')
elif mode == 'after_B_dead': leg('cfg-B'); leg('cfg-E')
elif mode == 'after_C_dead': leg('cfg-C'); leg('cfg-E')
elif mode == 'after_BCD': leg('cfg-B', keep=alive); leg('cfg-C'); leg('cfg-D'); leg('cfg-E'); leg('cfg-E', warmup=30)
elif mode == 'pool256': leg('cfg-E', pool_n=256)
