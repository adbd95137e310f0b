"""which handles of a process are slow?  FsmgModel directly (library-owned streams), device tokens via torch only for the pool"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np, torch, bench
from fsmg.binding import FsmgModel
mode = sys.argv[1]
cfg = dict(bench.CFG_B)
pool = bench.synthetic_episodes(8, 5, 5, 4, 128, 10000, seed=1)
d_sup = torch.from_numpy(np.stack([s for s, _ in pool])).cuda(); d_qry = torch.from_numpy(np.stack([q for _, q in pool])).cuda()
ss, qs = d_sup[0].numel() * 4, d_qry[0].numel() * 4
def run(m, tag):
    def step(i):
        e = i % 8; m.train_step(d_sup.data_ptr() + e * ss, d_qry.data_ptr() + e * qs, shape=(5, 5, 4), want_loss=False)
    for i in range(6): step(i)
    m.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(i)
    m.synchronize(); print('%s %s: %.4f ms/step' % (mode, tag, 1e3 * (time.perf_counter() - t0) / 20), flush=True)
keep = []
for k in range(1, 10):
    if mode == 'torch_streams':
        st = torch.cuda.Stream(); keep.append(st)
        m = FsmgModel(cfg, max_sequences=45, stream=st.cuda_stream)
    elif mode == 'torch_streams_x2':          # what HIPModel does: two torch streams per handle
        st = torch.cuda.Stream(); keep.append(st); keep.append(torch.cuda.Stream())
        m = FsmgModel(cfg, max_sequences=45, stream=st.cuda_stream)
    elif mode == 'destroy':                   # library-owned streams, handle destroyed before the next is created
        m = FsmgModel(cfg, max_sequences=45)
    else:                                     # 'own': library-owned streams, all alive
        m = FsmgModel(cfg, max_sequences=45); keep.append(m)
    m.init_params(1)
    run(m, 'handle %d' % k)
    if mode == 'destroy': m.close()
