#!/usr/bin/env python3
"""A bare cfg-B train loop for counter-collecting profiler passes (tools/pmc_passes.sh): no torch, no bench legs, host tokens.

  python tools/pmc_workload.py [steps]            env: FSMG_* select the kernel family / order as for any handle

Under `rocprofv3 --pmc` dispatches are serialised: the two launches of an XCD-partitioned pair cannot run side by side, the gated
projection times out (0.2 s) and the handle parks the order -- so with the default (AUTO) schedule the first passes of this loop run
the PACKED chain kernels (k_lstm_*_xcd16<4>) and the rest the same family chip-wide (k_lstm_*_xcd16<2>); FSMG_XCD_OVERLAP=0 gives the
fp32 kernels (k_lstm_*_xcd<2>).  Per-instantiation means come out of tools/pmc_to_json.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'))
import numpy as np                                     # noqa: E402
from fsmg.binding import FsmgModel                      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = dict(name='lstm_baseline', seed=1234, input_size=10000, max_len=128, embedding_size=250, hidden_size=512, n_layers=1,
           lr=5e-3, max_grad_norm=5, n_decay=10000)
rng = np.random.RandomState(1234)
eps = [(rng.randint(0, 10000, size=(5, 5, 128)).astype(np.int32), rng.randint(0, 10000, size=(5, 4, 128)).astype(np.int32)) for _ in range(4)]
m = FsmgModel(cfg, max_sequences=45)
m.init_params(1234)
m.debug_set('fallback_steps', 1)            # a timed-out pass is repeated once on per-step launches, then the persistent kernels are back
done = 0
for i in range(steps):
    try:
        loss = m.train_step(*eps[i % 4])
        done += 1
    except Exception as e:                  # noqa: BLE001 -- a time-out report: the step has been repeated or is repeated by the next call
        print('step %d: %s' % (i, str(e)[:160]), file=sys.stderr)
print('pmc_workload: %d of %d steps, global_step %d, stats %r' % (done, steps, m.step, m.stats()))
m.close()
