"""Episode-parallel path on CPU: world_size-2 gloo processes.

The HIP engine cannot run here, so the engine slot of fsmg.dist.EpisodeParallel is filled by a CPU stand-in
built on the oracle (test infrastructure) that exposes the same three members (forward_backward,
grad_tensor, apply_update) with the same flat-gradient + tail-scalars contract as libfsmg.  What is under
test is the product's host logic: the episode sharding, the ONE all-reduce per step, grad_scale = 1/world,
and that 2 ranks x 1 episode equal 1 rank on the 2 concatenated episodes.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import free_port, small_config
from oracle import lstm_oracle as O

CFG = small_config(hidden_size=12, embedding_size=6, input_size=40, max_len=6, max_grad_norm=0.5)
N, K, Q = 2, 2, 1
STEPS = 3


class OracleEngine(object):
    """CPU stand-in with libfsmg's engine contract: flat grad tensor whose tail holds [slices_sq, loss]."""
    TAIL = 16

    def __init__(self, cfg, params):
        self.cfg = cfg
        self.params = {k: v.copy() for k, v in params.items()}
        self.opt = O.new_opt_state(self.params)
        self.names = [n for n, _ in O.param_shapes(cfg)]
        self.sizes = [self.params[n].size for n in self.names]
        self.grad_tensor = torch.zeros(sum(self.sizes) + self.TAIL, dtype=torch.float64)
        # same bucket contract as the HIP model: [softmax_w + softmax_b | everything else | tail scalars]
        n_soft = self.sizes[-1] + self.sizes[-2]
        n_all = sum(self.sizes)
        self.grad_buckets = [self.grad_tensor[n_all - n_soft:n_all], self.grad_tensor[:n_all - n_soft],
                             self.grad_tensor[n_all:]]
        self.waited = []

    def comm_context(self, bucket):
        import contextlib
        self.waited.append(bucket)
        return contextlib.nullcontext()

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def forward_backward(self, support, query, **kw):
        X, Y = O.train_xy(support, query, self.cfg['input_size'])
        loss, cache = O.forward(self.params, X, Y, self.cfg)
        grads, aux = O.backward(self.params, cache, self.cfg)
        flat = np.concatenate([grads[n].ravel() for n in self.names] + [np.zeros(self.TAIL)])
        flat[-self.TAIL] = aux['embedding_slices_sq']
        flat[-self.TAIL + 1] = loss
        self.grad_tensor.copy_(torch.from_numpy(flat))

    def apply_update(self, grad_scale=1.0, want_loss=True):
        flat = self.grad_tensor.numpy() * 1.0
        grads, off = {}, 0
        for n, sz in zip(self.names, self.sizes):
            grads[n] = (flat[off:off + sz] * grad_scale).reshape(self.params[n].shape)
            off += sz
        aux = dict(embedding_slices_sq=flat[-self.TAIL] * grad_scale * grad_scale)
        O.apply_update(self.params, grads, aux, self.opt, self.cfg, 'tf1_slices')
        return float(flat[-self.TAIL + 1] * grad_scale)


def _episodes():
    return O.synthetic_episodes(2 * STEPS, N, K, Q, CFG['max_len'], CFG['input_size'], seed=3)


def _worker(rank, world, port, out_dir, bucketed='1'):
    os.environ['FSMG_DP_BUCKETS'] = bucketed
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'few-shot-music-generation_amd', 'src'))
    from fsmg.dist import EpisodeParallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    eng = OracleEngine(CFG, O.glorot_init(CFG, 5, np.float64))
    par = EpisodeParallel(eng)
    assert (par.rank, par.world) == (rank, world)
    eps = _episodes()
    losses = [par.train_step(*eps[s * world + rank]) for s in range(STEPS)]   # rank r takes episode s*R + r
    mean_val = par.mean_scalar(float(rank))
    assert eng.waited == ([0, 1, 2] * STEPS if bucketed == '1' else [])
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), losses=np.array(losses), mean_val=mean_val, **eng.params)
    dist.destroy_process_group()


@pytest.mark.parametrize('bucketed', ['1', '0'])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, bucketed):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), bucketed), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'rank1.npz'))
    # replicas stay identical and see the same (mean) loss
    np.testing.assert_array_equal(r0['losses'], r1['losses'])
    assert r0['mean_val'] == r1['mean_val'] == 0.5
    # single process on the 2-episode batch per step: loss = mean over all rows == mean of the two episode losses
    params = O.glorot_init(CFG, 5, np.float64)
    opt = O.new_opt_state(params)
    eps = _episodes()
    for s in range(STEPS):
        (s0, q0), (s1, q1) = eps[2 * s], eps[2 * s + 1]
        sup, qry = np.concatenate([s0, s1]), np.concatenate([q0, q1])
        want = O.train_step(params, opt, sup, qry, CFG)
        assert abs(r0['losses'][s] - want) <= 1e-12 * abs(want)
    for k, v in params.items():
        np.testing.assert_allclose(r0[k], v, rtol=1e-10, atol=1e-14, err_msg=k)
        np.testing.assert_array_equal(r0[k], r1[k])


def test_single_process_group_is_a_passthrough():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'few-shot-music-generation_amd', 'src'))
    from fsmg.dist import EpisodeParallel
    eng = OracleEngine(CFG, O.glorot_init(CFG, 5, np.float64))
    par = EpisodeParallel(eng)
    assert par.world == 1 and par.mean_scalar(3.0) == 3.0
    sup, qry = _episodes()[0]
    params = O.glorot_init(CFG, 5, np.float64)
    opt = O.new_opt_state(params)
    assert abs(par.train_step(sup, qry) - O.train_step(params, opt, sup, qry, CFG)) < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------
# R = 8 (the north star is 8 ranks; nothing had exercised the host logic beyond world 2)
class _TimeoutEngine(OracleEngine):
    """OracleEngine + libfsmg's time-out contract: a rank whose (simulated) persistent kernel timed out raises tail[2]; after the
    exchange EVERY rank sees the indicator, skips the update, reports it and has fallen back -- the driver repeats the step."""
    def __init__(self, cfg, params, timeout_at=()):
        super(_TimeoutEngine, self).__init__(cfg, params)
        self.timeout_at, self.calls, self.skipped = set(timeout_at), 0, 0

    def forward_backward(self, support, query, **kw):
        super(_TimeoutEngine, self).forward_backward(support, query, **kw)
        if self.calls in self.timeout_at:
            self.grad_tensor[-self.TAIL + 2] = 1.0           # this rank's gradients are garbage
            self.grad_tensor[:-self.TAIL] = float('nan')
        self.calls += 1

    def apply_update(self, grad_scale=1.0, want_loss=True):
        if float(self.grad_tensor[-self.TAIL + 2]) != 0.0:
            self.skipped += 1
            e = RuntimeError('FSMG_ERR_TIMEOUT: persistent recurrent kernel timed out waiting for a peer block')
            e.code = -9           # the status code is the contract (include/fsmg.h FSMG_ERR_TIMEOUT; fsmg.binding.FsmgError.code), not the text
            raise e
        return super(_TimeoutEngine, self).apply_update(grad_scale, want_loss)


class _StreamSampler(object):
    """an episode stream whose items are their own position (what ShardedEpisodeSampler wraps)"""
    def __init__(self):
        self.pos = 0

    def episode_indices(self):
        self.pos += 1
        return (self.pos - 1, None)

    def gather(self, a, b):
        return a


class _PosModel(object):
    def eval(self, episode):
        return float(episode) ** 2 + 1.0


def _worker8(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'few-shot-music-generation_amd', 'src'))
    from data.episode import ShardedEpisodeSampler
    from fsmg.dist import EpisodeParallel
    from train.train import sharded_validate
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FSMG_DP_BUCKETS='1')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # (a) one stream dealt round-robin over 8, then 3 steps with a time-out raised on rank 5 in step 1: all 8 skip, all 8 repeat
    eng = _TimeoutEngine(CFG, O.glorot_init(CFG, 5, np.float64), timeout_at=({1} if rank == 5 else ()))
    par = EpisodeParallel(eng)
    stream = O.synthetic_episodes(world * STEPS, N, K, Q, CFG['max_len'], CFG['input_size'], seed=13)
    losses = [par.train_step(*stream[s * world + rank]) for s in range(STEPS)]
    # (b) validation over n_val = 13 episodes (not divisible by 8), twice in a row: the stream copies must stay aligned
    sh = ShardedEpisodeSampler(_StreamSampler(), rank, world)
    v1 = sharded_validate(_PosModel(), sh, 13, rank, world, par)
    v2 = sharded_validate(_PosModel(), sh, 13, rank, world, par)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), losses=np.array(losses), skipped=eng.skipped, calls=eng.calls,
             v1=v1, v2=v2, pos=sh.sampler.pos, **eng.params)
    dist.destroy_process_group()


def test_eight_ranks_one_stream_timeout_on_one_rank_and_ragged_validation(tmp_path):
    world = 8
    port = free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]
    for r in rs[1:]:
        np.testing.assert_array_equal(r['losses'], rs[0]['losses'])              # replicas see the same mean loss ...
        for k in ('embedding', 'kernel_0', 'softmax_w'):
            np.testing.assert_array_equal(r[k], rs[0][k])                        # ... and stay bit-identical
    # the time-out of rank 5 in step 1 was seen by every rank (the indicator travels in the reduced tail): one skipped update each,
    # one repeated forward_backward each, in lock-step
    assert [int(r['skipped']) for r in rs] == [1] * world and [int(r['calls']) for r in rs] == [STEPS + 1] * world
    # 8 ranks x 1 episode == 1 rank on the 8-episode batch, three steps (the repeated step included once)
    params = O.glorot_init(CFG, 5, np.float64)
    opt = O.new_opt_state(params)
    stream = O.synthetic_episodes(world * STEPS, N, K, Q, CFG['max_len'], CFG['input_size'], seed=13)
    for s in range(STEPS):
        sup = np.concatenate([stream[s * world + r][0] for r in range(world)])
        qry = np.concatenate([stream[s * world + r][1] for r in range(world)])
        want = O.train_step(params, opt, sup, qry, CFG)
        assert abs(rs[0]['losses'][s] - want) <= 1e-12 * abs(want)
    for k, v in params.items():
        np.testing.assert_allclose(rs[0][k], v, rtol=1e-10, atol=1e-14, err_msg=k)
    # validation: the mean over EXACTLY the first 13 (then the next 13 after the alignment gap) stream positions
    want1 = np.mean([p ** 2 + 1.0 for p in range(13)])
    want2 = np.mean([p ** 2 + 1.0 for p in range(16, 29)])                      # every copy of the stream advanced by ceil(13 / 8) * 8 = 16
    for r in rs:
        assert abs(float(r['v1']) - want1) < 1e-9 and abs(float(r['v2']) - want2) < 1e-9
        assert int(r['pos']) == 32
