"""-m gpu: the episode-parallel path on the HIP engine.

  * 2 ranks, one episode each, equal one rank on the concatenated batch (and, for the MAML-style step, the oracle's mean of
    per-rank query gradients); replicas stay bit-identical.  Over RCCL when >= 2 GPUs are visible; on a 1-GPU box both
    ranks share the GPU and exchange over gloo;
  * on 1 GPU: the train step beside ANOTHER workload that holds CUs (large matmuls on a second stream -- what RCCL kernels or
    a neighbour job do to the persistent recurrent kernels' co-residency): same numbers, and a time-out, if one happens, is
    recovered and counted instead of poisoning later steps.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port, small_config
from gpu_utils import new_model
from oracle import lstm_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('hidden,maml,exchange', [(48, 0, 'torch'), (512, 0, 'torch'), (48, 1, 'torch'), (512, 0, 'library'), (48, 1, 'library')])
def test_two_ranks_on_the_hip_engine_equal_one_rank_on_the_concatenated_batch(hidden, maml, exchange, record_property, request):
    import re
    import warnings
    import torch
    if exchange == 'library' and torch.cuda.device_count() < 2:
        pytest.skip('the library-owned exchange is RCCL only: two ranks need two GPUs (this is the 2-GPU parity test the path is gated on)')
    if exchange == 'library':
        # the opt-in path (FSMG_ALLOW_LIBRARY_RCCL) has never run on two GPUs: its first run reports (xpassed / xfailed) instead of
        # stopping a `-x` suite; the default exchange (torch-issued RCCL) stays strict
        request.applymarker(pytest.mark.xfail(reason='first execution of the library-owned RCCL exchange on two GPUs', strict=False))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if torch.cuda.device_count() < 2:
        # RCCL cannot place two ranks on one device ("Duplicate GPU detected"): both ranks share GPU 0 and exchange over gloo --
        # the same host code (sharding, bucketed exchange on the communication stream, lock-step recovery when the two
        # processes' persistent kernels get in each other's way); the RCCL leg runs wherever 2 GPUs are visible
        env['FSMG_TEST_SAME_GPU'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_hip_worker.py'), str(hidden), str(maml), exchange]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600, env=env)
    assert proc.returncode == 0 and 'DIST_HIP_OK' in proc.stdout, proc.stdout[-4000:]
    # say which transport carried the exchange: a 1-GPU box cannot run RCCL with two ranks, and a green test must not read as
    # "RCCL verified" there
    m = re.search(r'backend=(\w+) devices=(\d+)', proc.stdout)
    backend, ndev = (m.group(1), int(m.group(2))) if m else ('?', 0)
    record_property('exchange_backend', backend)
    record_property('distinct_gpus', ndev)
    print('episode-parallel HIP test ran over backend=%s on %d GPU(s)' % (backend, ndev))
    if backend != 'nccl':
        warnings.warn('2-rank HIP-engine test exchanged gradients over %s on ONE GPU: RCCL (backend nccl) was NOT exercised '
                      '(needs >= 2 visible GPUs)' % backend)


@pytest.mark.parametrize('order', ['serial', 'xcd_partitioned'])
def test_train_steps_beside_a_cu_hogging_workload_keep_their_numbers(order, monkeypatch):
    """What a collective kernel of an overlapped exchange does to the step: another stream keeps every CU busy with matmuls while train
    steps run.  The XCD-local chains (and, order = xcd_partitioned, the gated pair of chain and work-queue GEMM, which additionally needs
    its two launches resident side by side) either run as usual -- same bits -- or time out, fall back and repeat the step: never a hang,
    never a wrong number."""
    import torch
    monkeypatch.setenv('FSMG_XCD_OVERLAP', '1' if order == 'xcd_partitioned' else '0')
    cfg = small_config(hidden_size=512, embedding_size=64, input_size=2000, max_len=32)
    eps = O.synthetic_episodes(12, 5, 5, 4, cfg['max_len'], cfg['input_size'], seed=51)
    quiet = new_model(cfg)
    want = [quiet.train_step(s, q) for s, q in eps]
    busy = new_model(cfg)
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device='cuda')
    b = torch.randn(8192, 8192, device='cuda')
    torch.cuda.synchronize()
    got = []
    with torch.cuda.stream(side):
        for _ in range(6):
            c = a @ b                                    # ~7 ms each on every CU: the train steps below run beside them
    for s, q in eps:
        got.append(busy.train_step(s, q))
        with torch.cuda.stream(side):
            c = a @ b
    torch.cuda.synchronize()
    stats = busy.stats()
    assert busy.step == len(eps) and stats['steps_skipped_token_range'] == 0
    assert int(quiet.debug_read('xcd_partitioned', 3)[2]) == (1 if order == 'xcd_partitioned' else 0)
    if stats['timeouts'] == 0:
        assert got == want                               # same kernels, same order: same bits
    else:                                                # a recovered time-out repeats the step on per-step launches
        for g, w in zip(got, want):
            assert abs(g - w) <= 1e-5 * abs(w)
    del c
