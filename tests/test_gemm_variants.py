"""The bf16-split GEMM has several implementations of the same arithmetic (csrc/gemm.hip): k_gemm_bx3 (128 x 128 tiles), its
wave-specialised variant k_gemm_bx3w and the 256 x 256-tile k_gemm_bx3h (chosen per shape by use_ws_gemm / use_h_gemm,
csrc/api_schedule.hip) and, in all of them, operand loads as buffer loads or through 64-bit lane addresses (k_gemm_bx3h: two
x-contiguous operands through LDS-DMA as well, FSMG_GEMM_DMA).  DESIGN.md claims they produce THE SAME BITS for the same K split -- same LDS image, k order and term order --
which is what lets the choice be made per shape by measured speed alone.  ("For the same K split": the variants keep a different
number of blocks per CU, so the split policy may cut K differently for them and the slabs are then summed in a different
association; FSMG_MAX_SPLIT=1 takes that out of the comparison.)  The knobs are read once per process, so every variant
runs in its own interpreter on the same seeded episodes at a production-shaped problem (hidden 512, vocabulary 10001, 45 rows,
so that every GEMM class of the step takes part) and reports a digest of the losses and of every parameter after three updates."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, json, sys
import numpy as np
sys.path[:0] = [%(tests)r, %(root)r, %(src)r]
from gpu_utils import new_model
from oracle import lstm_oracle as O
cfg = dict(input_size=10000, embedding_size=250, hidden_size=512, n_layers=1, max_len=24, max_grad_norm=5.0, lr=1e-3,
           n_decay=1000, seed=3)
eps = O.synthetic_episodes(3, 5, 5, 4, cfg['max_len'], cfg['input_size'], seed=5, realistic=True)
m = new_model(cfg)
losses = [m.train_step(s, q) for s, q in eps]
h = hashlib.sha256()
for name in sorted(m.param_shapes):
    h.update(np.ascontiguousarray(m.get_param(name)).tobytes())
print(json.dumps({'losses': [float(x) for x in losses], 'params_sha256': h.hexdigest()}))
'''


def _run(**env_over):
    env = dict(os.environ)
    env['FSMG_MAX_SPLIT'] = '1'             # one K range per GEMM for every variant
    env.update(env_over)
    proc = subprocess.run([sys.executable, '-c', SCRIPT % {'tests': os.path.join(ROOT, 'tests'), 'root': ROOT, 'src': os.path.join(ROOT, 'few-shot-music-generation_amd', 'src')}], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return json.loads(proc.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_every_bf16_split_gemm_variant_gives_the_same_bits():
    # FSMG_FUSED_SOFTMAX=0: the fused softmax of a train pass exists in the 256 x 256-tile kernel only and is a
    # different (tolerance-equal, not bit-equal) arithmetic of the cross entropy -- with it on, WHICH softmax runs would follow the
    # GEMM variant; its own variants are compared below
    classic = dict(FSMG_FUSED_SOFTMAX='0')
    base = _run(FSMG_GEMM_WS='0', FSMG_GEMM_H='0', FSMG_GEMM_BUF='0', **classic)
    assert all(x == x and x > 0 for x in base['losses'])
    for over in (dict(FSMG_GEMM_WS='2', FSMG_GEMM_H='0', FSMG_GEMM_BUF='0'), dict(FSMG_GEMM_WS='0', FSMG_GEMM_H='0', FSMG_GEMM_BUF='1'),
                 dict(FSMG_GEMM_WS='2', FSMG_GEMM_H='0', FSMG_GEMM_BUF='1'), dict(FSMG_GEMM_H='2', FSMG_GEMM_BUF='0'),
                 dict(FSMG_GEMM_H='2', FSMG_GEMM_BUF='1'), dict(FSMG_GEMM_H='2', FSMG_GEMM_DMA='0'),
                 dict(FSMG_GEMM_H='2', FSMG_MERGE_DK='0'),      # dKx and dKh as two GEMMs instead of one with a two-part A (GemmArgs::m_split)
                 dict()):
        got = _run(**dict(over, **classic))
        assert got == base, (over, got, base)


@pytest.mark.gpu
def test_the_fused_softmax_gives_the_same_bits_in_every_variant_of_the_256_tile_kernel():
    """exp(logit) + row sums from the projection's epilogue, weighted column sums in dW (one dword load + DPP row broadcasts; LDS-DMA or
    register staging), row-scaled dH sum: buffer loads or lane addresses, LDS-DMA or not -- one set of bits; and those bits are
    the classic softmax's within the tolerance the parity tests state (here: losses to 1e-5 relative)."""
    base = _run(FSMG_GEMM_H='2', FSMG_GEMM_BUF='1')
    for over in (dict(FSMG_GEMM_H='2', FSMG_GEMM_BUF='0'), dict(FSMG_GEMM_H='2', FSMG_GEMM_DMA='0'), dict(FSMG_GEMM_H='2', FSMG_MERGE_DK='0')):
        got = _run(**over)
        assert got == base, (over, got, base)
    classic = _run(FSMG_GEMM_H='2', FSMG_GEMM_BUF='1', FSMG_FUSED_SOFTMAX='0')
    assert classic['params_sha256'] != base['params_sha256'], 'the fused softmax did not run'
    for a, b in zip(base['losses'], classic['losses']):
        assert abs(a - b) <= 1e-5 * abs(b), (base['losses'], classic['losses'])
