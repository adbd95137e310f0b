import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/few-shot-music-generation_amd/src')
import numpy as np, torch
from oracle import lstm_oracle as O
from oracle.torch_ref import TorchRef
import bench
cfg = bench.CFG_B
pool = O.synthetic_episodes(4, 5, 5, 4, cfg['max_len'], cfg['input_size'], seed=1234)
for th in (8, 16, 32, 64, 128):
    ref = TorchRef(cfg, O.glorot_init(cfg, 1234, np.float32), dtype=torch.float32, threads=th)
    ref.train(*pool[0])
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 8 and n < 6:
        ref.train(*pool[n % 4]); n += 1
    dt = time.perf_counter() - t0
    print('threads %d: %.3f episodes/s (%d steps %.1fs)' % (th, n / dt, n, dt), flush=True)
