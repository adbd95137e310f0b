"""Worker of tests/test_dist_hip.py: one rank of a 2-GPU episode-parallel job on the HIP engine over RCCL.
Launched as `python -m torch.distributed.run --nproc-per-node 2 tests/_dist_hip_worker.py <hidden> <maml>`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'few-shot-music-generation_amd', 'src'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

from conftest import small_config       # noqa: E402
from fsmg.binding import FsmgModel      # noqa: E402
from fsmg.dist import init_from_env     # noqa: E402
from oracle import lstm_oracle as O     # noqa: E402


def main():
    hidden, maml = int(sys.argv[1]), int(sys.argv[2])
    exchange = sys.argv[3] if len(sys.argv) > 3 else 'torch'          # 'library': the RCCL calls issued by libfsmg (fsmg_comm_*)
    # FSMG_TEST_SAME_GPU=1: both ranks on GPU 0 with the gloo backend (RCCL refuses two ranks on one device) -- exercises the
    # same host code (sharding, bucketed exchange on the communication stream, lock-step recovery) on a 1-GPU box
    same_gpu = os.environ.get('FSMG_TEST_SAME_GPU', '0') == '1'
    rank, world = init_from_env('gloo' if same_gpu else 'nccl')
    local = 0 if same_gpu else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    cfg = small_config(hidden_size=hidden, embedding_size=24, input_size=150, max_len=10, max_grad_norm=0.5, device=local,
                       name='maml_lstm' if maml else 'lstm_baseline', inner_steps=1, inner_lr=0.2, dp_exchange=exchange)
    if exchange == 'library':
        os.environ['FSMG_ALLOW_LIBRARY_RCCL'] = '1'
    if same_gpu:
        os.environ['LOCAL_RANK'] = '0'
    if maml:
        from models.maml_lstm import MAMLLSTM as Model
    else:
        from models.lstm_baseline import LSTMBaseline as Model
    model = Model(dict(cfg))
    model.recover_or_init('')                      # rank 0's seeded init is broadcast over RCCL
    start = model.engine.get_params()
    eps = O.synthetic_episodes(3 * world, 3, 2, 2, cfg['max_len'], cfg['input_size'], seed=77, realistic=True)
    from data.episode import Episode
    losses = [model.train(Episode(*eps[s * world + rank])) for s in range(3)]
    got = model.engine.get_params()
    stats = model.engine.stats()
    assert same_gpu or stats['timeouts'] == 0, stats      # two processes time-slicing one GPU may time out and recover
    # every rank holds the same parameters, bit for bit
    flat = torch.from_numpy(np.concatenate([v.ravel() for v in got.values()])).cuda()
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref), 'replicas diverged'
    if rank == 0:
        if maml:
            # oracle: mean over the ranks' first-order query gradients, one clip + Adam per outer step
            params = {k: v.astype(np.float64) for k, v in start.items()}
            opt = O.new_opt_state(params)
            for s in range(3):
                acc, slices, loss = None, 0.0, 0.0
                for r in range(world):
                    l, g, aux = O.maml_query_grads(params, *eps[s * world + r], cfg, 1, 0.2)
                    acc = g if acc is None else {k: acc[k] + g[k] for k in g}
                    slices += aux['embedding_slices_sq']
                    loss += l
                O.apply_update(params, {k: v / world for k, v in acc.items()}, {'embedding_slices_sq': slices / world ** 2}, opt, cfg)
                assert abs(losses[s] - loss / world) <= 1e-4 * abs(loss / world), (s, losses[s], loss / world)
            for k, v in params.items():
                assert np.abs(got[k] - v).max() <= 5e-4 * max(np.abs(v).max(), 1e-6), k
        else:
            # one rank on the concatenated batch: the same update
            single = FsmgModel(cfg, device=local)
            single.init_params(0)
            single.set_params(start)
            for s in range(3):
                sup = np.concatenate([eps[s * world + r][0] for r in range(world)])
                qry = np.concatenate([eps[s * world + r][1] for r in range(world)])
                want = single.train_step(sup, qry)
                assert abs(losses[s] - want) <= 1e-5 * abs(want), (s, losses[s], want)
            for k, v in single.get_params().items():
                # Adam's first steps are sign-like (m / sqrt(v)): a 1e-7 difference in a tiny gradient (another summation
                # order: 2 x 15 rows vs 30 rows pick different kernels) moves a weight by a fraction of lr
                assert np.abs(got[k] - v).max() <= 2e-4 * max(np.abs(v).max(), 1e-6), k
        print('DIST_HIP_OK world=%d hidden=%d maml=%d backend=%s devices=%d' % (world, hidden, maml, dist.get_backend(),
                                                                                  1 if same_gpu else world))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
