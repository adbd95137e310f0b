#!/usr/bin/env python3
"""Phase timeline of the recurrent step kernels at cfg-B (s_memtime stamps, instrumented build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import bench
from fsmg.binding import FsmgModel
cfg = dict(bench.CFG_B)
m = FsmgModel(cfg, use_graph=False); m.init_params(1)
(sup, qry), = bench.synthetic_episodes(1, 5, 5, 4, cfg['max_len'], cfg['input_size'], 1234)
for which, name in ((0, 'fwd'), (1, 'bwd')):
    m.forward_backward(sup, qry)
    st = m.step_profile(which).astype(np.int64)
    t0 = st[:, :, 0].min()
    rel = st - t0
    print('%s: %d blocks x %d waves; stamps relative to the first wave entry (s_memtime ticks)' % (name, st.shape[0], st.shape[1]))
    for i, lab in enumerate(['entry', 'loads landed', 'partials in LDS', 'past barrier', 'done']):
        v = rel[:, :, i]
        print('  %-16s min %7d  median %7d  max %7d' % (lab, v.min(), int(np.median(v)), v.max()))
    d = st[:, :, 1:5] - st[:, :, 0:4]
    print('  per-wave phase medians: load %d  mfma+lds %d  barrier %d  epilogue %d' % tuple(int(np.median(d[:, :, i])) for i in range(4)))
    print('  wave0 epilogue median %d, kernel span (last done - first entry) %d' % (int(np.median(d[:, 0, 3])), rel[:, :, 4].max()))
