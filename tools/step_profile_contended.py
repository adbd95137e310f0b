#!/usr/bin/env python3
"""Phase timeline of the recurrent step kernels at cfg-B, alone and while an MFMA-bound fp32 GEMM (torch.matmul on a
side stream) or an HBM-bound copy co-runs: which phase of a step does contention stretch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np
import torch
import bench
from fsmg.binding import FsmgModel
cfg = dict(bench.CFG_B)
m = FsmgModel(cfg, use_graph=False); m.init_params(1)
(sup, qry), = bench.synthetic_episodes(1, 5, 5, 4, cfg['max_len'], cfg['input_size'], 1234)
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device='cuda'); b = torch.randn(8192, 8192, device='cuda'); c = torch.empty_like(a)
big = torch.empty(1 << 28, device='cuda'); big2 = torch.empty_like(big)
torch.cuda.synchronize()

def background(kind):
    with torch.cuda.stream(side):
        if kind == 'gemm':
            for _ in range(6): torch.matmul(a, b, out=c)
        elif kind == 'copy':
            for _ in range(40): big2.copy_(big)

for kind in ('alone', 'gemm', 'copy'):
    for which, name in ((0, 'fwd'), (1, 'bwd')):
        m.forward_backward(sup, qry)
        m.synchronize() if hasattr(m, 'synchronize') else torch.cuda.synchronize()
        if kind != 'alone': background(kind)
        st = m.step_profile(which).astype(np.int64)
        torch.cuda.synchronize()
        d = st[:, :, 1:5] - st[:, :, 0:4]
        span = st[:, :, 4].max(axis=1) - st[:, :, 0].min(axis=1)       # per block (one XCD clock domain per block)
        print('%-5s %s: per-wave phase medians: load %5d  mfma+lds %5d  barrier %5d  epilogue %5d | p90: load %5d mfma %5d barrier %5d epi %5d | block span median %5d max %5d' % (
            kind, name, *[int(np.median(d[:, :, i])) for i in range(4)], *[int(np.percentile(d[:, :, i], 90)) for i in range(4)],
            int(np.median(span)), int(span.max())))
