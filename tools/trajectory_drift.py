import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/few-shot-music-generation_amd/src'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import small_config
from gpu_utils import new_model, f64_params
from oracle import lstm_oracle as O
cfg = small_config(hidden_size=48, embedding_size=24, input_size=301, max_len=16, n_layers=1)
model = new_model(cfg)
params = f64_params(model); opt = O.new_opt_state(params)
p32 = {k: v.astype(np.float32) for k, v in params.items()}; o32 = O.new_opt_state(p32)
eps = O.synthetic_episodes(60, 3, 3, 2, cfg['max_len'], cfg['input_size'], seed=4, realistic=True)
worst = 0; worst32 = 0
for s, (sup, qry) in enumerate(eps):
    a = O.train_step(params, opt, sup, qry, cfg); b = model.train_step(sup, qry); c = O.train_step(p32, o32, sup, qry, cfg)
    worst = max(worst, abs(a - b) / abs(a)); worst32 = max(worst32, abs(a - c) / abs(a))
    if s % 10 == 9: print('step %2d oracle64 %.6f hip %.6f numpy32 %.6f  worst rel so far hip %.2e numpy-fp32 %.2e' % (s + 1, a, b, c, worst, worst32))
